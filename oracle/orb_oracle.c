/*
 * orb_oracle.c — CPU restatement of the ORB front end (FAST-9/16 + intensity-centroid orientation +
 * 256-bit steered BRIEF on a 1.2x / 8-level pyramid).  TEST INFRASTRUCTURE ONLY (see oracle/README).
 *
 * PARITY UNPINNED: the reference tree contains no ORB extractor (it lives in the un-vendored external
 * plugin pi-gslam/gslam_orbslam -> raulmur/ORB_SLAM + OpenCV, no version pinned: README.md:131,
 * doc/doxygen/4_1_orbslam.dox:7) and none of its tests hold a golden descriptor.  What the reference
 * does pin are the OUTPUT types, which this file honours:
 *   GSLAM/core/Map.h:122-195   KeyPoint {pt.x, pt.y, size, angle, response, octave, class_id} 28 B
 *   GSLAM/core/Map.h:309-321   MapFrame::setKeyPoints(keypoints, descriptors = N x 32 8UC1 GImage)
 *   GSLAM/core/Vocabulary.h:485-491 reads a descriptor as 4 little-endian u64 -> bit k of the test
 *                              vector is stored LSB-first in byte k/8.
 * The algorithm follows the published ORB / ORB-SLAM ORBextractor design (Rublee et al. 2011;
 * Mur-Artal et al. 2015) made integer end to end so that CPU and GPU agree bit for bit
 * (SURVEY.md 8c).  The numbered steps below are the specification; DESIGN.md repeats it.
 *
 *  1 pyramid  level l is w_l x h_l, w_l = round_half_up(W * 5^l / 6^l) (exact integer); level l is
 *             resized from level l-1, bilinear, 11-bit fixed-point weights:
 *             P = floor((2x+1) * w_src * 2048 / (2 w_dst)) - 1024, clamped at 0; sx = P >> 11,
 *             fx = P & 2047 (sx >= w_src-1 -> sx = w_src-1, fx = 0); same for y;
 *             v = (sum of 4 taps * weights + 2^21) >> 22.
 *  2 FAST     score(x,y) = max over the 16 arcs of 9 contiguous ring pixels of min(ring - p) and of
 *             min(p - ring); a pixel is a corner at threshold t iff score > t.  Only pixels with
 *             19 <= x < w-19, 19 <= y < h-19 are scored; S = score if score > min_th else 0.
 *  3 NMS      candidate iff S > 0 and S > S(neighbour) for all 8 neighbours (strict; outside = 0).
 *  4 cells    32x32 cells anchored at (19,19).  A cell holding a candidate with S > ini_th keeps only
 *             candidates with S > ini_th, else all.  In-cell rank by (S desc, y asc, x asc); ranks
 *             >= 32 are dropped.
 *  5 select   per-level quota n_l = round_half_up(K 5^l 6^(L-1-l) / (6^L - 5^L)) clamped so the running
 *             total never exceeds K, last level takes the remainder.  Take the first n_l candidates of the level under the total order
 *             (rank asc, S desc, cell index asc, in-cell raster asc).  Output order: level asc,
 *             cell index asc (row-major), in-cell raster (y asc, x asc).
 *  6 angle    m10 = sum u I, m01 = sum v I over the radius-15 disc (GH_ORB_UMAX); orientation bin k
 *             in 0..29 is the unique k with cross(dir[k-1], m) >= 0 and cross(dir[k], m) < 0
 *             (GH_ORB_DIR, 64-bit integers); m == 0 -> bin 0.  angle = 12 k degrees.
 *  7 blur     B = (sum_{i,j in -3..3} g_i g_j I + 2^21) >> 22, GH_ORB_GAUSS taps (sum 2048).
 *  8 BRIEF    bit k = B(p + a_k) < B(p + b_k) with (a_k, b_k) = GH_ORB_PATTERN[bin][k]; byte k/8,
 *             bit k%8 (LSB first).
 *  9 keypoint pt = (float)x_l * s_l, s_0 = 1, s_l = s_{l-1} * 1.2f (single fp32 roundings);
 *             size = 31 * s_l; response = S; octave = l; class_id = -1.
 *
 * Quadtree distribution (oracle_orb_set_distribution(1) <-> gh_orb_plan_set_distribution(plan, 1)) replaces steps 3-5 by what
 * ORB-SLAM's ORBextractor does there (ComputeKeyPointsOctTree + DistributeOctTree + ExtractorNode::DivideNode; not in the
 * reference tree, restated from the published algorithm; "border" below = EDGE_THRESHOLD - 3 = 16):
 *  4' cells   W' = w - 32, H' = h - 32; nCols = max(1, W' / 30), nRows = max(1, H' / 30) (integer division),
 *             wCell = ceil(W' / nCols), hCell = ceil(H' / nRows).  ORB-SLAM runs OpenCV FAST on the window of cell (i, j) that
 *             starts at (16 + j wCell, 16 + i hCell) and is wCell + 6 x hCell + 6 (clipped at w - 16, h - 16); FAST leaves a 3 px
 *             border, so cell (i, j) DETECTS in [19 + j wCell, min(19 + (j + 1) wCell, w - 19)) x (same in y): the cells tile
 *             the scored region of step 2.  A pixel is a candidate of its cell at threshold t iff S > t and S > S(neighbour)
 *             for the neighbours that lie INSIDE the cell's detection region (OpenCV's non-maximum suppression never sees
 *             the others).  t = ini_th; a cell without a candidate at ini_th is searched again at min_th.  No per-cell cap.
 *  5' tree    per level, keys (x - 16, y - 16) in the region W' x H', N = n_l (the quota of step 5):
 *             nIni = max(1, round(W' / H')) (fp32 division, half away from zero), hX = (float)W' / nIni (fp32); root i spans
 *             x in [(int)(hX i), (int)(hX (i + 1))) x [0, H'); a key joins root min((int)(kx / hX), nIni - 1).  Empty roots
 *             are dropped.  DivideNode: halfX = ceil((x1 - x0) / 2), halfY = ceil((y1 - y0) / 2), the four children are
 *             [x0, x0 + halfX) / [x0 + halfX, x1) x [y0, y0 + halfY) / [y0 + halfY, y1); a key goes left iff
 *             kx < x0 + halfX, up iff ky < y0 + halfY; empty children are dropped.
 *             Loop: (A) split EVERY node that holds more than one key; let n = nodes, e = children of this pass holding
 *             more than one key.  If n >= N or n did not change: stop.  If n + 3 e > N: (B) take the nodes created by the
 *             last pass that hold more than one key in the order (keys desc, y0 asc, x0 asc) [ORB-SLAM: std::sort of
 *             (size, pointer) pairs walked from the back -- its tie order is the heap's; this one is fixed] and split them
 *             one by one until n >= N; if the list is exhausted and n changed, repeat (B) with the children just made;
 *             stop when n >= N or a whole pass leaves n unchanged.  Otherwise (A) again.
 *             Each node yields its key of greatest S (ties: smaller y, then smaller x).  The tree can end with up to 3 nodes more
 *             than N (ORB-SLAM keeps them all; the output here has n_l rows per level): the n_l best by (S desc, y asc, x asc)
 *             are kept.  Output order: level asc, then y asc, x asc.
 *
 * Continuous steering (oracle_orb_set_steer(1) <-> gh_orb_plan_set_steering(plan, 1)) replaces steps 6 and 8 by what
 * OpenCV / ORB-SLAM do there (ORBextractor.cc IC_Angle + computeOrbDescriptor; neither is in the reference tree, see above):
 *  6' angle   a = fastAtan2((float)m01, (float)m10) in degrees -- OpenCV's fp32 polynomial: with ax = |x|, ay = |y|,
 *             c = min / (max + (float)DBL_EPSILON), a = (((p7 c^2 + p5) c^2 + p3) c^2 + p1) c (or 90 - that when ay > ax),
 *             a = 180 - a if x < 0, a = 360 - a if y < 0; p1, p3, p5, p7 = 57.283627f, -18.667446f, 8.9140005f, -2.5397246f.
 *             Every operation a single fp32 rounding, no contraction.  KeyPoint.angle = a.
 *  8' BRIEF   (cos, sin)(a): k = (int)(a / 90 + 0.5), r = (a - 90 k) * 0.017453292f, Taylor polynomials of degree 7 / 8 in r
 *             (fp32, Horner, coefficients below), quadrant fixed up by k & 3.  Test point (x, y) of the UNROTATED pattern ->
 *             (x', y') = (rintf(x cos - y sin), rintf(x sin + y cos)) (ties to even = cvRound);
 *             bit k = B(p + a'_k) < B(p + b'_k).  B is the blur of step 7 over the level with BORDER_REFLECT_101
 *             coordinates (index -i -> i, n - 1 + i -> n - 1 - i): points reach 19 px, the blur 22, the border is 19 away.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "gslam_orb_tables.h"

typedef struct oracle_kp {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} oracle_kp;

static int64_t ipow(int b, int e) {
  int64_t r = 1;
  while (e-- > 0) r *= b;
  return r;
}

void oracle_orb_level_dims(int w, int h, int nlevels, int* ws, int* hs) {
  for (int l = 0; l < nlevels; ++l) {
    int64_t p5 = ipow(5, l), p6 = ipow(6, l);
    ws[l] = (int)((2 * (int64_t)w * p5 + p6) / (2 * p6));
    hs[l] = (int)((2 * (int64_t)h * p5 + p6) / (2 * p6));
  }
}

void oracle_orb_quotas(int K, int nlevels, int* q) {
  int64_t den = ipow(6, nlevels) - ipow(5, nlevels);
  int sum = 0;
  for (int l = 0; l < nlevels - 1; ++l) {
    int64_t num = (int64_t)K * ipow(5, l) * ipow(6, nlevels - 1 - l);
    q[l] = (int)((2 * num + den) / (2 * den));
    if (q[l] > K - sum) q[l] = K - sum; /* rounding may not push the total past K */
    sum += q[l];
  }
  q[nlevels - 1] = K - sum > 0 ? K - sum : 0;
}

void oracle_orb_scales(int nlevels, float* s) {
  s[0] = 1.0f;
  for (int l = 1; l < nlevels; ++l) s[l] = s[l - 1] * 1.2f;
}

/* step 1 */
static void resize_axis_table(int n_src, int n_dst, int* idx, int* frac) {
  for (int x = 0; x < n_dst; ++x) {
    int64_t P = ((int64_t)(2 * x + 1) * n_src * 2048) / (2 * (int64_t)n_dst) - 1024;
    if (P < 0) P = 0;
    int sx = (int)(P >> 11), fx = (int)(P & 2047);
    if (sx >= n_src - 1) {
      sx = n_src - 1;
      fx = 0;
    }
    idx[x] = sx;
    frac[x] = fx;
  }
}

void oracle_orb_resize(const uint8_t* src, int ws, int hs, int sstride, uint8_t* dst, int wd, int hd, int dstride) {
  int* xi = (int*)malloc(sizeof(int) * wd * 2);
  int* yi = (int*)malloc(sizeof(int) * hd * 2);
  resize_axis_table(ws, wd, xi, xi + wd);
  resize_axis_table(hs, hd, yi, yi + hd);
  for (int y = 0; y < hd; ++y) {
    int sy = yi[y], fy = yi[hd + y];
    int sy1 = sy + 1 < hs ? sy + 1 : hs - 1;
    const uint8_t* r0 = src + (size_t)sy * sstride;
    const uint8_t* r1 = src + (size_t)sy1 * sstride;
    for (int x = 0; x < wd; ++x) {
      int sx = xi[x], fx = xi[wd + x];
      int sx1 = sx + 1 < ws ? sx + 1 : ws - 1;
      uint32_t v = (uint32_t)r0[sx] * (2048 - fx) * (2048 - fy) + (uint32_t)r0[sx1] * fx * (2048 - fy) +
                   (uint32_t)r1[sx] * (2048 - fx) * fy + (uint32_t)r1[sx1] * fx * fy;
      dst[(size_t)y * dstride + x] = (uint8_t)((v + (1u << 21)) >> 22);
    }
  }
  free(xi);
  free(yi);
}

/* step 2: FAST corner score of the pixel at p (ring offsets precomputed for the stride) */
static int fast_score(const uint8_t* p, const int* ring_off) {
  int d[16 + 8];
  int c = *p;
  for (int i = 0; i < 16; ++i) d[i] = (int)p[ring_off[i]] - c;
  for (int i = 0; i < 8; ++i) d[16 + i] = d[i];
  int best = 0;
  for (int a = 0; a < 16; ++a) {
    int mn = d[a], mx = d[a];
    for (int i = 1; i < 9; ++i) {
      if (d[a + i] < mn) mn = d[a + i];
      if (d[a + i] > mx) mx = d[a + i];
    }
    if (mn > best) best = mn;   /* all 9 brighter by at least mn */
    if (-mx > best) best = -mx; /* all 9 darker by at least -mx */
  }
  return best;
}

/* Score map S of one level (steps 2): 0 outside the valid region or when score <= min_th. */
void oracle_orb_score_map(const uint8_t* img, int w, int h, int stride, int min_th, uint8_t* S) {
  int ring_off[16];
  for (int i = 0; i < 16; ++i) ring_off[i] = GH_ORB_RING[i][1] * stride + GH_ORB_RING[i][0];
  memset(S, 0, (size_t)w * h);
  for (int y = GH_ORB_EDGE; y < h - GH_ORB_EDGE; ++y)
    for (int x = GH_ORB_EDGE; x < w - GH_ORB_EDGE; ++x) {
      const uint8_t* p = img + (size_t)y * stride + x;
      /* cheap necessary condition (any 9-arc holds >= 2 of the 4 compass pixels) - exact, not a spec change */
      int c = *p, nb = 0, nd = 0;
      int r0 = p[ring_off[0]], r4 = p[ring_off[4]], r8 = p[ring_off[8]], r12 = p[ring_off[12]];
      nb = (r0 > c + min_th) + (r4 > c + min_th) + (r8 > c + min_th) + (r12 > c + min_th);
      nd = (r0 < c - min_th) + (r4 < c - min_th) + (r8 < c - min_th) + (r12 < c - min_th);
      if (nb < 2 && nd < 2) continue;
      int s = fast_score(p, ring_off);
      if (s > min_th) S[(size_t)y * w + x] = (uint8_t)(s > 255 ? 255 : s);
    }
}

typedef struct {
  int x, y, s, rank, cell, order; /* order = position in (cell asc, in-cell raster) traversal */
} cand_t;

static int cmp_cell_rank(const void* a, const void* b) { /* (S desc, y asc, x asc) */
  const cand_t* p = (const cand_t*)a;
  const cand_t* q = (const cand_t*)b;
  if (p->s != q->s) return q->s - p->s;
  if (p->y != q->y) return p->y - q->y;
  return p->x - q->x;
}
static int cmp_raster(const void* a, const void* b) {
  const cand_t* p = (const cand_t*)a;
  const cand_t* q = (const cand_t*)b;
  if (p->y != q->y) return p->y - q->y;
  return p->x - q->x;
}
static int cmp_select(const void* a, const void* b) { /* (rank asc, S desc, order asc) */
  const cand_t* p = (const cand_t*)a;
  const cand_t* q = (const cand_t*)b;
  if (p->rank != q->rank) return p->rank - q->rank;
  if (p->s != q->s) return q->s - p->s;
  return p->order - q->order;
}
static int cmp_order(const void* a, const void* b) {
  return ((const cand_t*)a)->order - ((const cand_t*)b)->order;
}

/* steps 3-5 for one level.  Returns number selected, written in output order. */
int oracle_orb_select_level(const uint8_t* S, int w, int h, int ini_th, int quota, int* out_x, int* out_y,
                            int* out_s) {
  int vw = w - 2 * GH_ORB_EDGE, vh = h - 2 * GH_ORB_EDGE;
  if (vw <= 0 || vh <= 0 || quota <= 0) return 0;
  int ncx = (vw + GH_ORB_CELL - 1) / GH_ORB_CELL, ncy = (vh + GH_ORB_CELL - 1) / GH_ORB_CELL;
  cand_t* all = (cand_t*)malloc(sizeof(cand_t) * ((size_t)vw * vh / 4 + 16));
  cand_t cellbuf[GH_ORB_CELL * GH_ORB_CELL / 4 + 4];
  int n_all = 0;
  for (int cy = 0; cy < ncy; ++cy)
    for (int cx = 0; cx < ncx; ++cx) {
      int n = 0, strong = 0;
      int x0 = GH_ORB_EDGE + cx * GH_ORB_CELL, y0 = GH_ORB_EDGE + cy * GH_ORB_CELL;
      for (int y = y0; y < y0 + GH_ORB_CELL && y < h - GH_ORB_EDGE; ++y)
        for (int x = x0; x < x0 + GH_ORB_CELL && x < w - GH_ORB_EDGE; ++x) {
          int s = S[(size_t)y * w + x];
          if (s == 0) continue;
          int ismax = 1;
          for (int dy = -1; dy <= 1 && ismax; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
              if (!dx && !dy) continue;
              if (S[(size_t)(y + dy) * w + (x + dx)] >= s) {
                ismax = 0;
                break;
              }
            }
          if (!ismax) continue;
          cellbuf[n].x = x;
          cellbuf[n].y = y;
          cellbuf[n].s = s;
          cellbuf[n].cell = cy * ncx + cx;
          ++n;
          if (s > ini_th) strong = 1;
        }
      if (strong) {
        int m = 0;
        for (int i = 0; i < n; ++i)
          if (cellbuf[i].s > ini_th) cellbuf[m++] = cellbuf[i];
        n = m;
      }
      qsort(cellbuf, n, sizeof(cand_t), cmp_cell_rank);
      if (n > GH_ORB_CELL_CAP) n = GH_ORB_CELL_CAP;
      for (int i = 0; i < n; ++i) cellbuf[i].rank = i;
      qsort(cellbuf, n, sizeof(cand_t), cmp_raster);
      for (int i = 0; i < n; ++i) {
        cellbuf[i].order = n_all;
        all[n_all++] = cellbuf[i];
      }
    }
  qsort(all, n_all, sizeof(cand_t), cmp_select);
  int n_sel = n_all < quota ? n_all : quota;
  qsort(all, n_sel, sizeof(cand_t), cmp_order);
  for (int i = 0; i < n_sel; ++i) {
    out_x[i] = all[i].x;
    out_y[i] = all[i].y;
    out_s[i] = all[i].s;
  }
  free(all);
  return n_sel;
}

/* ---- quadtree distribution (steps 4' and 5') */
static int g_distribution = 0;
void oracle_orb_set_distribution(int mode) { g_distribution = mode != 0; }

void oracle_orb_slam_grid(int w, int h, int* ncols, int* nrows, int* wcell, int* hcell) {
  const int W2 = w - 32, H2 = h - 32;
  *ncols = W2 / 30 > 1 ? W2 / 30 : 1;
  *nrows = H2 / 30 > 1 ? H2 / 30 : 1;
  *wcell = (W2 + *ncols - 1) / *ncols;
  *hcell = (H2 + *nrows - 1) / *nrows;
}

/* step 4': candidates of one level in (cell row-major, in-cell raster) order; returns their number */
int oracle_orb_slam_candidates(const uint8_t* S, int w, int h, int ini_th, int* cx, int* cy, int* cs) {
  int ncols, nrows, wc, hc, n = 0;
  if (w <= 2 * GH_ORB_EDGE || h <= 2 * GH_ORB_EDGE) return 0;
  oracle_orb_slam_grid(w, h, &ncols, &nrows, &wc, &hc);
  for (int i = 0; i < nrows; ++i)
    for (int j = 0; j < ncols; ++j) {
      const int x0 = GH_ORB_EDGE + j * wc, y0 = GH_ORB_EDGE + i * hc;
      const int x1 = x0 + wc < w - GH_ORB_EDGE ? x0 + wc : w - GH_ORB_EDGE;
      const int y1 = y0 + hc < h - GH_ORB_EDGE ? y0 + hc : h - GH_ORB_EDGE;
      const int first = n;
      int strong = 0;
      for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
          const int s = S[(size_t)y * w + x];
          if (s == 0) continue;
          int ismax = 1;
          for (int dy = -1; dy <= 1 && ismax; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
              const int xx = x + dx, yy = y + dy;
              if ((!dx && !dy) || xx < x0 || xx >= x1 || yy < y0 || yy >= y1) continue;
              if (S[(size_t)yy * w + xx] >= s) {
                ismax = 0;
                break;
              }
            }
          if (!ismax) continue;
          cx[n] = x;
          cy[n] = y;
          cs[n] = s;
          ++n;
          if (s > ini_th) strong = 1;
        }
      if (strong) {
        int m = first;
        for (int k = first; k < n; ++k)
          if (cs[k] > ini_th) {
            cx[m] = cx[k];
            cy[m] = cy[k];
            cs[m] = cs[k];
            ++m;
          }
        n = m;
      }
    }
  return n;
}

typedef struct qt_node {
  int x0, y0, x1, y1; /* [x0, x1) x [y0, y1) in region coordinates */
  int* keys;          /* indices into the candidate arrays, in arrival order */
  int n;
  struct qt_node *prev, *next;
} qt_node;

typedef struct {
  qt_node *head, *tail;
  int size;
} qt_list;

static qt_node* qt_new(int x0, int y0, int x1, int y1, int cap) {
  qt_node* q = (qt_node*)calloc(1, sizeof(qt_node));
  q->x0 = x0; q->y0 = y0; q->x1 = x1; q->y1 = y1;
  q->keys = (int*)malloc(sizeof(int) * (cap > 0 ? cap : 1));
  return q;
}
static void qt_push_front(qt_list* L, qt_node* q) {
  q->prev = NULL;
  q->next = L->head;
  if (L->head) L->head->prev = q; else L->tail = q;
  L->head = q;
  ++L->size;
}
static void qt_push_back(qt_list* L, qt_node* q) {
  q->next = NULL;
  q->prev = L->tail;
  if (L->tail) L->tail->next = q; else L->head = q;
  L->tail = q;
  ++L->size;
}
static void qt_erase(qt_list* L, qt_node* q) {
  if (q->prev) q->prev->next = q->next; else L->head = q->next;
  if (q->next) q->next->prev = q->prev; else L->tail = q->prev;
  --L->size;
  free(q->keys);
  free(q);
}
/* DivideNode; children with keys go to the front of the list, those with more than one key also into `made` */
static void qt_divide(qt_list* L, qt_node* q, const int* kx, const int* ky, qt_node** made, int* n_made) {
  const int hx = (q->x1 - q->x0 + 1) / 2, hy = (q->y1 - q->y0 + 1) / 2; /* ceil of the half extent */
  const int xm = q->x0 + hx, ym = q->y0 + hy;
  qt_node* c[4] = {qt_new(q->x0, q->y0, xm, ym, q->n), qt_new(xm, q->y0, q->x1, ym, q->n), qt_new(q->x0, ym, xm, q->y1, q->n),
                   qt_new(xm, ym, q->x1, q->y1, q->n)};
  for (int k = 0; k < q->n; ++k) {
    const int id = q->keys[k];
    qt_node* t = kx[id] < xm ? (ky[id] < ym ? c[0] : c[2]) : (ky[id] < ym ? c[1] : c[3]);
    t->keys[t->n++] = id;
  }
  for (int e = 0; e < 4; ++e) {
    if (c[e]->n > 0) {
      qt_push_front(L, c[e]);
      if (c[e]->n > 1) made[(*n_made)++] = c[e];
    } else {
      free(c[e]->keys);
      free(c[e]);
    }
  }
}
static int qt_cmp_expand(const void* a, const void* b) { /* (keys desc, y0 asc, x0 asc) */
  const qt_node* p = *(qt_node* const*)a;
  const qt_node* q = *(qt_node* const*)b;
  if (p->n != q->n) return q->n - p->n;
  if (p->y0 != q->y0) return p->y0 - q->y0;
  return p->x0 - q->x0;
}
typedef struct { int x, y, s; } qt_win;
static int qt_cmp_best(const void* a, const void* b) { /* (S desc, y asc, x asc) */
  const qt_win* p = (const qt_win*)a;
  const qt_win* q = (const qt_win*)b;
  if (p->s != q->s) return q->s - p->s;
  if (p->y != q->y) return p->y - q->y;
  return p->x - q->x;
}
static int qt_cmp_yx(const void* a, const void* b) {
  const qt_win* p = (const qt_win*)a;
  const qt_win* q = (const qt_win*)b;
  if (p->y != q->y) return p->y - q->y;
  return p->x - q->x;
}

/* step 5' for one level: candidates (level coordinates) -> at most N keypoints in (y, x) order.  Returns their number. */
int oracle_orb_quadtree(const int* cx, const int* cy, const int* cs, int n, int w, int h, int N, int* out_x, int* out_y,
                        int* out_s) {
  if (n <= 0 || N <= 0) return 0;
  const int W2 = w - 32, H2 = h - 32;
  int* kx = (int*)malloc(sizeof(int) * n * 2);
  int* ky = kx + n;
  for (int k = 0; k < n; ++k) {
    kx[k] = cx[k] - 16;
    ky[k] = cy[k] - 16;
  }
  int n_ini = (int)roundf((float)W2 / (float)H2);
  if (n_ini < 1) n_ini = 1;
  const float hX = (float)W2 / (float)n_ini;
  qt_list L = {NULL, NULL, 0};
  qt_node** roots = (qt_node**)malloc(sizeof(qt_node*) * n_ini);
  for (int i = 0; i < n_ini; ++i) {
    roots[i] = qt_new((int)(hX * (float)i), 0, (int)(hX * (float)(i + 1)), H2, n);
    qt_push_back(&L, roots[i]);
  }
  for (int k = 0; k < n; ++k) {
    int r = (int)((float)kx[k] / hX);
    if (r > n_ini - 1) r = n_ini - 1;
    roots[r]->keys[roots[r]->n++] = k;
  }
  for (int i = 0; i < n_ini; ++i)
    if (roots[i]->n == 0) qt_erase(&L, roots[i]);
  free(roots);
  qt_node** made = (qt_node**)malloc(sizeof(qt_node*) * ((size_t)4 * n + 16));
  qt_node** prev = (qt_node**)malloc(sizeof(qt_node*) * ((size_t)4 * n + 16));
  int finish = 0;
  while (!finish) {
    const int prev_size = L.size;
    int n_made = 0;
    for (qt_node* q = L.head; q;) { /* (A): children go to the FRONT, so this walk meets only nodes of earlier passes */
      qt_node* nx = q->next;
      if (q->n > 1) {
        qt_divide(&L, q, kx, ky, made, &n_made);
        qt_erase(&L, q);
      }
      q = nx;
    }
    if (L.size >= N || L.size == prev_size) {
      finish = 1;
    } else if (L.size + 3 * n_made > N) {
      while (!finish) {
        const int before = L.size;
        const int n_prev = n_made;
        memcpy(prev, made, sizeof(qt_node*) * n_prev);
        n_made = 0;
        qsort(prev, n_prev, sizeof(qt_node*), qt_cmp_expand);
        for (int j = 0; j < n_prev; ++j) {
          qt_divide(&L, prev[j], kx, ky, made, &n_made);
          qt_erase(&L, prev[j]);
          if (L.size >= N) break;
        }
        if (L.size >= N || L.size == before) finish = 1;
      }
    }
  }
  qt_win* win = (qt_win*)malloc(sizeof(qt_win) * (L.size > 0 ? L.size : 1));
  int m = 0;
  for (qt_node* q = L.head; q; q = q->next) {
    qt_win b = {0, 0, -1};
    for (int k = 0; k < q->n; ++k) {
      const qt_win c = {cx[q->keys[k]], cy[q->keys[k]], cs[q->keys[k]]};
      if (b.s < 0 || qt_cmp_best(&c, &b) < 0) b = c;
    }
    win[m++] = b;
  }
  while (L.head) qt_erase(&L, L.head);
  qsort(win, m, sizeof(qt_win), qt_cmp_best);
  if (m > N) m = N;
  qsort(win, m, sizeof(qt_win), qt_cmp_yx);
  for (int i = 0; i < m; ++i) {
    out_x[i] = win[i].x;
    out_y[i] = win[i].y;
    out_s[i] = win[i].s;
  }
  free(win);
  free(made);
  free(prev);
  free(kx);
  return m;
}

/* steps 4' + 5' for one level */
int oracle_orb_select_level_quadtree(const uint8_t* S, int w, int h, int ini_th, int quota, int* out_x, int* out_y,
                                     int* out_s) {
  if (w <= 2 * GH_ORB_EDGE || h <= 2 * GH_ORB_EDGE || quota <= 0) return 0;
  const size_t cap = (size_t)w * h / 2 + 64;
  int* cx = (int*)malloc(sizeof(int) * cap * 3);
  int* cy = cx + cap;
  int* cs = cy + cap;
  const int n = oracle_orb_slam_candidates(S, w, h, ini_th, cx, cy, cs);
  const int m = oracle_orb_quadtree(cx, cy, cs, n, w, h, quota, out_x, out_y, out_s);
  free(cx);
  return m;
}

/* step 6 */
int oracle_orb_angle_bin(const uint8_t* img, int stride, int x, int y) {
  int64_t m10 = 0, m01 = 0;
  for (int v = -GH_ORB_HALF_PATCH; v <= GH_ORB_HALF_PATCH; ++v) {
    int um = GH_ORB_UMAX[v < 0 ? -v : v];
    const uint8_t* row = img + (size_t)(y + v) * stride + x;
    for (int u = -um; u <= um; ++u) {
      m10 += (int64_t)u * row[u];
      m01 += (int64_t)v * row[u];
    }
  }
  if (m10 == 0 && m01 == 0) return 0;
  for (int k = 0; k < GH_ORB_NBINS; ++k) {
    int km = (k + GH_ORB_NBINS - 1) % GH_ORB_NBINS;
    int64_t c_lo = (int64_t)GH_ORB_DIR[km][0] * m01 - (int64_t)GH_ORB_DIR[km][1] * m10;
    int64_t c_hi = (int64_t)GH_ORB_DIR[k][0] * m01 - (int64_t)GH_ORB_DIR[k][1] * m10;
    if (c_lo >= 0 && c_hi < 0) return k;
  }
  return 0;
}

/* step 7 */
static inline int blur_at(const uint8_t* img, int stride, int x, int y) {
  uint32_t acc = 0;
  for (int j = -3; j <= 3; ++j) {
    const uint8_t* row = img + (size_t)(y + j) * stride + x;
    uint32_t h = 0;
    for (int i = -3; i <= 3; ++i) h += (uint32_t)GH_ORB_GAUSS[i + 3] * row[i];
    acc += (uint32_t)GH_ORB_GAUSS[j + 3] * h;
  }
  return (int)((acc + (1u << 21)) >> 22);
}

/* ---- continuous steering (steps 6' and 8') */
static int g_steer = 0;
void oracle_orb_set_steer(int mode) { g_steer = mode != 0; }

float oracle_orb_fast_atan2_deg(float y, float x) {
  const float p1 = 57.283627f, p3 = -18.667446f, p5 = 8.9140005f, p7 = -2.5397246f;
  const float eps = 2.2204460492503131e-16f;
  const float ax = fabsf(x), ay = fabsf(y);
  float a;
  if (ax >= ay) {
    const float c = ay / (ax + eps), c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    const float c = ax / (ay + eps), c2 = c * c;
    a = 90.0f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0.0f) a = 180.0f - a;
  if (y < 0.0f) a = 360.0f - a;
  return a;
}

void oracle_orb_sincos_deg(float a, float* cs, float* sn) {
  const int k = (int)(a / 90.0f + 0.5f);
  const float r = (a - 90.0f * (float)k) * 0.017453292f, r2 = r * r;
  const float s = r * (1.0f + r2 * (-0.16666667f + r2 * (0.0083333338f + r2 * -0.00019841270f)));
  const float c = 1.0f + r2 * (-0.5f + r2 * (0.041666668f + r2 * (-0.0013888889f + r2 * 0.000024801588f)));
  switch (k & 3) {
    case 0: *cs = c; *sn = s; break;
    case 1: *cs = -s; *sn = c; break;
    case 2: *cs = -c; *sn = -s; break;
    default: *cs = s; *sn = -c; break;
  }
}

static inline int reflect101(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

static inline int blur_at_reflect(const uint8_t* img, int w, int h, int stride, int x, int y) {
  uint32_t acc = 0;
  for (int j = -3; j <= 3; ++j) {
    const uint8_t* row = img + (size_t)reflect101(y + j, h) * stride;
    uint32_t hs = 0;
    for (int i = -3; i <= 3; ++i) hs += (uint32_t)GH_ORB_GAUSS[i + 3] * row[reflect101(x + i, w)];
    acc += (uint32_t)GH_ORB_GAUSS[j + 3] * hs;
  }
  return (int)((acc + (1u << 21)) >> 22);
}

float oracle_orb_angle_deg(const uint8_t* img, int stride, int x, int y) {
  int64_t m10 = 0, m01 = 0;
  for (int v = -GH_ORB_HALF_PATCH; v <= GH_ORB_HALF_PATCH; ++v) {
    int um = GH_ORB_UMAX[v < 0 ? -v : v];
    const uint8_t* row = img + (size_t)(y + v) * stride + x;
    for (int u = -um; u <= um; ++u) {
      m10 += (int64_t)u * row[u];
      m01 += (int64_t)v * row[u];
    }
  }
  return oracle_orb_fast_atan2_deg((float)m01, (float)m10);
}

/* Test pattern in use: the built-in table, or one installed by oracle_orb_set_pattern (the checker's counterpart of
 * gh_orb_plan_set_pattern: same rotation rule as tools/gen_orb_tables.py -- round half away from zero).  Global state:
 * tests install / reset it around single-threaded runs. */
static int8_t g_custom_pattern[GH_ORB_NBINS][256][4];
static int8_t g_custom_base[256][4];
static int g_use_custom_pattern = 0;

/* Returns -1 when the pattern cannot be used in the current steering mode: a point beyond radius 19.49, or (30-bin mode)
 * a rotated coordinate beyond +-13. */
int oracle_orb_set_pattern(const int8_t* base /* 256 x 4, NULL = back to the built-in pattern */) {
  if (!base) {
    g_use_custom_pattern = 0;
    return 0;
  }
  for (int t = 0; t < 256; ++t)
    for (int e = 0; e < 4; e += 2)
      if ((int)base[4 * t + e] * base[4 * t + e] + (int)base[4 * t + e + 1] * base[4 * t + e + 1] > 379) return -1;
  int fits = 1;
  for (int k = 0; k < GH_ORB_NBINS; ++k) {
    const double th = (12.0 * k) * (3.14159265358979323846 / 180.0), c = cos(th), s = sin(th);
    for (int t = 0; t < 256; ++t) {
      const int8_t* q = base + 4 * t;
      const double v[4] = {q[0] * c - q[1] * s, q[0] * s + q[1] * c, q[2] * c - q[3] * s, q[2] * s + q[3] * c};
      for (int e = 0; e < 4; ++e) {
        const int r = (int)floor(fabs(v[e]) + 0.5) * (v[e] >= 0 ? 1 : -1);
        if (r < -13 || r > 13) fits = 0;
        g_custom_pattern[k][t][e] = (int8_t)(r < -13 ? -13 : (r > 13 ? 13 : r));
      }
    }
  }
  if (!fits && !g_steer) return -1;
  memcpy(g_custom_base, base, sizeof(g_custom_base));
  g_use_custom_pattern = 1;
  return 0;
}

/* step 8' */
void oracle_orb_describe_steer(const uint8_t* img, int w, int h, int stride, int x, int y, float angle, uint8_t* desc32) {
  float cs, sn;
  oracle_orb_sincos_deg(angle, &cs, &sn);
  memset(desc32, 0, 32);
  for (int k = 0; k < 256; ++k) {
    const int8_t* p = g_use_custom_pattern ? g_custom_base[k] : GH_ORB_PATTERN[0][k];
    const float ax = (float)p[0], ay = (float)p[1], bx = (float)p[2], by = (float)p[3];
    const int rax = (int)rintf(ax * cs - ay * sn), ray = (int)rintf(ax * sn + ay * cs);
    const int rbx = (int)rintf(bx * cs - by * sn), rby = (int)rintf(bx * sn + by * cs);
    const int a = blur_at_reflect(img, w, h, stride, x + rax, y + ray);
    const int b = blur_at_reflect(img, w, h, stride, x + rbx, y + rby);
    if (a < b) desc32[k >> 3] |= (uint8_t)(1u << (k & 7));
  }
}

/* step 8 */
void oracle_orb_describe(const uint8_t* img, int stride, int x, int y, int bin, uint8_t* desc32) {
  memset(desc32, 0, 32);
  for (int k = 0; k < 256; ++k) {
    const int8_t* p = g_use_custom_pattern ? g_custom_pattern[bin][k] : GH_ORB_PATTERN[bin][k];
    int a = blur_at(img, stride, x + p[0], y + p[1]);
    int b = blur_at(img, stride, x + p[2], y + p[3]);
    if (a < b) desc32[k >> 3] |= (uint8_t)(1u << (k & 7));
  }
}

/* Whole pipeline.  Returns the number of keypoints (<= K).  If pyr_out != NULL it receives pointers to the
 * malloc'ed level images (caller frees), for debugging. */
int oracle_orb_extract(const uint8_t* gray, int w, int h, int stride, int K, int nlevels, int ini_th, int min_th,
                       oracle_kp* kps, uint8_t* desc) {
  int ws[GH_ORB_MAX_LEVELS], hs[GH_ORB_MAX_LEVELS], quota[GH_ORB_MAX_LEVELS];
  float scale[GH_ORB_MAX_LEVELS];
  if (nlevels < 1 || nlevels > GH_ORB_MAX_LEVELS) return -1;
  oracle_orb_level_dims(w, h, nlevels, ws, hs);
  oracle_orb_quotas(K, nlevels, quota);
  oracle_orb_scales(nlevels, scale);
  uint8_t* lv[GH_ORB_MAX_LEVELS];
  int ls[GH_ORB_MAX_LEVELS];
  lv[0] = (uint8_t*)gray;
  ls[0] = stride;
  for (int l = 1; l < nlevels; ++l) {
    lv[l] = (uint8_t*)malloc((size_t)ws[l] * hs[l]);
    ls[l] = ws[l];
    oracle_orb_resize(lv[l - 1], ws[l - 1], hs[l - 1], ls[l - 1], lv[l], ws[l], hs[l], ls[l]);
  }
  int n = 0;
  for (int l = 0; l < nlevels; ++l) {
    if (quota[l] <= 0 || ws[l] <= 2 * GH_ORB_EDGE || hs[l] <= 2 * GH_ORB_EDGE) continue;
    uint8_t* S = (uint8_t*)malloc((size_t)ws[l] * hs[l]);
    oracle_orb_score_map(lv[l], ws[l], hs[l], ls[l], min_th, S);
    int* xs = (int*)malloc(sizeof(int) * quota[l] * 3);
    int* ys = xs + quota[l];
    int* ss = ys + quota[l];
    int m = g_distribution ? oracle_orb_select_level_quadtree(S, ws[l], hs[l], ini_th, quota[l], xs, ys, ss)
                           : oracle_orb_select_level(S, ws[l], hs[l], ini_th, quota[l], xs, ys, ss);
    for (int i = 0; i < m; ++i) {
      oracle_kp* kp = &kps[n];
      kp->x = (float)xs[i] * scale[l];
      kp->y = (float)ys[i] * scale[l];
      kp->size = 31.0f * scale[l];
      kp->response = (float)ss[i];
      kp->octave = l;
      kp->class_id = -1;
      if (g_steer) {
        kp->angle = oracle_orb_angle_deg(lv[l], ls[l], xs[i], ys[i]);
        oracle_orb_describe_steer(lv[l], ws[l], hs[l], ls[l], xs[i], ys[i], kp->angle, desc + (size_t)n * 32);
      } else {
        int bin = oracle_orb_angle_bin(lv[l], ls[l], xs[i], ys[i]);
        kp->angle = 12.0f * (float)bin;
        oracle_orb_describe(lv[l], ls[l], xs[i], ys[i], bin, desc + (size_t)n * 32);
      }
      ++n;
    }
    free(xs);
    free(S);
  }
  for (int l = 1; l < nlevels; ++l) free(lv[l]);
  return n;
}

/* Debug: write pyramid level `level` (dense w_l x h_l). */
int oracle_orb_pyramid_level(const uint8_t* gray, int w, int h, int stride, int nlevels, int level, uint8_t* out) {
  int ws[GH_ORB_MAX_LEVELS], hs[GH_ORB_MAX_LEVELS];
  oracle_orb_level_dims(w, h, nlevels, ws, hs);
  uint8_t* prev = (uint8_t*)gray;
  int ps = stride;
  for (int l = 1; l <= level; ++l) {
    uint8_t* cur = (uint8_t*)malloc((size_t)ws[l] * hs[l]);
    oracle_orb_resize(prev, ws[l - 1], hs[l - 1], ps, cur, ws[l], hs[l], ws[l]);
    if (l > 1) free(prev);
    prev = cur;
    ps = ws[l];
  }
  for (int y = 0; y < hs[level]; ++y) memcpy(out + (size_t)y * ws[level], prev + (size_t)y * ps, ws[level]);
  if (level > 0) free(prev);
  return 0;
}

/* Batch wrapper for the timed CPU baseline: frames are independent -> OpenMP over frames. */
void oracle_orb_extract_batch(const uint8_t* gray, int nframes, size_t frame_stride, int w, int h, int stride, int K,
                              int nlevels, int ini_th, int min_th, oracle_kp* kps, uint8_t* desc, int32_t* counts,
                              int threads) {
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
  for (int f = 0; f < nframes; ++f)
    counts[f] = oracle_orb_extract(gray + (size_t)f * frame_stride, w, h, stride, K, nlevels, ini_th, min_th,
                                   kps + (size_t)f * K, desc + (size_t)f * K * 32);
}

void oracle_bgr_to_gray(const uint8_t* bgr, int w, int h, int channels, int sstride, uint8_t* gray, int dstride) {
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const uint8_t* p = bgr + (size_t)y * sstride + (size_t)x * channels;
      gray[(size_t)y * dstride + x] = (uint8_t)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14);
    }
}
