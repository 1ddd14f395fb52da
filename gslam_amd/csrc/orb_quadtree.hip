// ORB-SLAM style keypoint distribution for the ORB front end (gh_orb_plan_set_distribution(plan, 1)): FAST per ~30 x 30
// cell with the cell's own threshold fallback and non-maximum suppression, then ORBextractor's DistributeOctTree per
// (frame, level).  Bit-exact with oracle/orb_oracle.c steps 4' and 5', which are the specification (ORB-SLAM's extractor
// is not part of the reference tree: README.md:131, doc/doxygen/4_1_orbslam.dox:7; the outputs feed the same
// GSLAM/core/Map.h:122-195 KeyPoint rows as the default mode).
//
// CDNA4 mapping.
//   slam_cells     one WAVE per cell (cells up to 32 x 32: every level but the smallest ones; four cells per workgroup, no
//                  block-wide barrier): tile + 3-px ring in LDS by dword loads, compass test on every pixel (exact necessary
//                  condition), the ~11 % survivors compacted by ballots, the 16-arc score only on those with every lane
//                  busy, 8-neighbour suppression inside the cell on a zero-bordered score plane, candidates appended to the
//                  (frame, level) key list (order is irrelevant: every later step is a function of the key SET).  Larger
//                  cells (a level narrower than 2 x 30 px is ONE column of cells up to 59 px wide) take the plain
//                  one-workgroup-per-cell kernel.
//   slam_quadtree  one 1024-lane workgroup per (frame, level).  ORB-SLAM's list of nodes becomes a level-synchronous
//                  table in LDS: a pass = one sweep over the keys counting the four children of every node being split
//                  (LDS atomics), one block scan that renumbers the table, one sweep that moves the keys.  The "largest
//                  nodes first until N" stage is a bitonic sort of the candidates + a scan of their gains, so the whole
//                  tree costs ~2 sweeps per generation instead of a pointer chase.
#include "orb_quadtree.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <new>

#include "../../include/gslam_orb_tables.h"

namespace {

constexpr int kEdge = GH_ORB_EDGE;
constexpr int kMaxL = GH_ORB_MAX_LEVELS;
constexpr int kBorder = kEdge - 3;      // ORB-SLAM's minBorder: the key coordinates of the tree start here
constexpr int kQtNodes = 2048;          // table capacity: quota + 3 and 4 * roots must fit
constexpr int kQtThreads = 1024;
constexpr int kCellMax = 59;            // wCell = ceil(W' / floor(W' / 30)) < 60
constexpr int kCellPitch = 68;          // (kCellMax + 6) bytes per tile row, padded
constexpr int kCellList = 30 * 30;      // suppressed maxima of a cell: at most ceil(59 / 2)^2

struct QtLevel {
  int w, h, quota, quota_off, n_ini;
  float hx;
  int ncols, nrows, wc, hc;
  uint32_t cap;       // key slots of this level per frame
  size_t key_off;     // first slot of this level inside a frame's block
};

struct QtArgs {
  QtLevel lv[kMaxL];
};

// the radius-3 ring in circular order (== GH_ORB_RING, checked in gh_qt_create)
struct Ring16 { int dx[16], dy[16]; };
constexpr Ring16 kRing = {{0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1}, {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3}};

// FAST-9/16 score of oracle step 2 at q (tile pitch `pitch`): the largest t such that 9 contiguous ring pixels are all
// brighter than centre + t - 1 or all darker than centre - t + 1, as max over the 16 arcs of min / -max of the differences.
__device__ __forceinline__ int fast_score_px(const uint8_t* q, int pitch, int min_th) {
  const int c = q[0];
  int d[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) d[i] = (int)q[kRing.dy[i] * pitch + kRing.dx[i]] - c;
  // necessary condition (every 9-arc holds at least two of the four compass points): exact, skips most pixels
  const int nb = (d[0] > min_th) + (d[4] > min_th) + (d[8] > min_th) + (d[12] > min_th);
  const int nd = (d[0] < -min_th) + (d[4] < -min_th) + (d[8] < -min_th) + (d[12] < -min_th);
  if (nb < 2 && nd < 2) return 0;
  // min / max over every window of 9 by doubling: windows of 2, 4, 8, then one more element (4 x 16 ops instead of 8 x 16)
  int lo[16], hi[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    lo[i] = min(d[i], d[(i + 1) & 15]);
    hi[i] = max(d[i], d[(i + 1) & 15]);
  }
  int lo4[16], hi4[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    lo4[i] = min(lo[i], lo[(i + 2) & 15]);
    hi4[i] = max(hi[i], hi[(i + 2) & 15]);
  }
  int best = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int mn = min(min(lo4[i], lo4[(i + 4) & 15]), d[(i + 8) & 15]);
    const int mx = max(max(hi4[i], hi4[(i + 4) & 15]), d[(i + 8) & 15]);
    best = max(best, max(mn, -mx));
  }
  return best;
}

// The two halves of fast_score_px for the wave-per-cell kernel: the compass test, and the arc score of a survivor.
__device__ __forceinline__ bool fast_compass_px(const uint8_t* q, int pitch, int min_th) {
  const int c = q[0];
  const int d0 = (int)q[-3 * pitch] - c, d4 = (int)q[3] - c, d8 = (int)q[3 * pitch] - c, d12 = (int)q[-3] - c;
  const int nb = (d0 > min_th) + (d4 > min_th) + (d8 > min_th) + (d12 > min_th);
  const int nd = (d0 < -min_th) + (d4 < -min_th) + (d8 < -min_th) + (d12 < -min_th);
  return nb >= 2 || nd >= 2;
}
__device__ __forceinline__ int fast_arc_score_px(const uint8_t* q, int pitch) {
  const int c = q[0];
  int d[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) d[i] = (int)q[kRing.dy[i] * pitch + kRing.dx[i]] - c;
  int lo[16], hi[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    lo[i] = min(d[i], d[(i + 1) & 15]);
    hi[i] = max(d[i], d[(i + 1) & 15]);
  }
  int lo4[16], hi4[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    lo4[i] = min(lo[i], lo[(i + 2) & 15]);
    hi4[i] = max(hi[i], hi[(i + 2) & 15]);
  }
  int best = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int mn = min(min(lo4[i], lo4[(i + 4) & 15]), d[(i + 8) & 15]);
    const int mx = max(max(hi4[i], hi4[(i + 4) & 15]), d[(i + 8) & 15]);
    best = max(best, max(mn, -mx));
  }
  return best;
}

// step 4', cells up to kWcMax x kWcMax: one wave per cell, four cells per workgroup (independent: no block-wide barrier)
constexpr int kWcMax = 32;
constexpr int kWRowDw = 11;                    // dwords per tile row: (3 bytes of alignment + 32 + 6 + 3) / 4
constexpr int kWImgPitch = 4 * kWRowDw;        // 44
constexpr int kWSPitch = 36;                   // score plane (kWcMax + 2)^2 with a zero border, padded rows
struct WaveCellLds {
  uint32_t img[(kWcMax + 6) * kWRowDw];        // 1672 B; dead after the arc scores: the list of suppressed maxima (at most
                                               // 16 x 16 words) takes its place
  uint32_t S[(kWcMax + 2) * kWSPitch / 4];     // 1224 B
  uint16_t queue[kWcMax * kWcMax];             // compass survivors, (y << 5) | x: 2048 B
};
static_assert(sizeof(WaveCellLds) * 4 <= 20 * 1024, "eight workgroups per CU");
static_assert((kWcMax / 2) * (kWcMax / 2) <= (kWcMax + 6) * kWRowDw, "the maxima fit where the tile was");

__global__ __launch_bounds__(256, 8) void slam_cells_wave_kernel(LevelView lv, int ncols, int ncells, int wc, int hc, int min_th,
                                                              int ini_th, uint32_t* __restrict__ keys, size_t keys_per_frame,
                                                              uint32_t cap, uint32_t* __restrict__ key_cnt, int level,
                                                              uint32_t* __restrict__ flags) {
  __shared__ WaveCellLds sh[4];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, b = blockIdx.y;
  const int cell = (int)blockIdx.x * 4 + wv;
  if (cell >= ncells) return;  // (wave-uniform)
  WaveCellLds& L = sh[wv];
  uint32_t* list = L.img;
  const int ci = cell / ncols, cj = cell - ci * ncols;
  const int x0 = kEdge + cj * wc, y0 = kEdge + ci * hc;
  const int x1 = min(x0 + wc, lv.w - kEdge), y1 = min(y0 + hc, lv.h - kEdge);
  const int cw = x1 - x0, ch = y1 - y0;
  if (cw <= 0 || ch <= 0) return;
  const uint8_t* img = lv.base + (size_t)b * lv.frame_stride;
  // tile rows y0 - 3 .. y1 + 2 as the aligned dwords that cover columns x0 - 3 .. x1 + 2 (the level's pitch is a multiple of
  // 4 and rows are padded to it: checked at the launch); pixel (r, c) of the tile is byte r * 44 + al + c
  const int xa = (x0 - 3) & ~3, al = (x0 - 3) & 3;
  const int ndw = (al + cw + 6 + 3) >> 2, nrow = ch + 6;
  for (int idx = lane; idx < nrow * kWRowDw; idx += 64) {
    // idx / 11 on the 24-bit multiplier (full rate; a division by a constant costs a quarter-rate v_mul_hi_u32): exact for
    // idx < 418 since 5958 / 65536 - 1 / 11 = 2.8e-6
    const int r = (int)(__umul24((uint32_t)idx, 5958u) >> 16), d = idx - r * kWRowDw;
    // (rows and pitch are < 2^24, a level is < 4 GiB: 32-bit offset on the full-rate 24-bit multiplier)
    if (d < ndw) L.img[idx] = *reinterpret_cast<const uint32_t*>(img + (__umul24((uint32_t)(y0 - 3 + r), (uint32_t)lv.pitch) + (uint32_t)(xa + 4 * d)));
  }
  for (int idx = lane; idx < (int)(sizeof(L.S) / 4); idx += 64) L.S[idx] = 0u;
  __builtin_amdgcn_wave_barrier();
  const uint8_t* I = reinterpret_cast<const uint8_t*>(L.img) + 3 * kWImgPitch + al + 3;  // pixel (0, 0) of the cell
  uint8_t* S = reinterpret_cast<uint8_t*>(L.S) + kWSPitch + 1;                            // score of cell pixel (0, 0)
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  // compass test on every pixel, survivors queued
  const int npx = cw * ch;
  const uint32_t inv = (65536u + (uint32_t)cw - 1u) / (uint32_t)cw;  // p / cw == (p * inv) >> 16 for p < 1024, cw <= 32
  int nq = 0;
  for (int base = 0; base < npx; base += 64) {
    const int p = base + lane;
    bool pass = false;
    uint32_t yx = 0;
    if (p < npx) {
      const int y = (int)(__umul24((uint32_t)p, inv) >> 16), x = p - (int)__umul24((uint32_t)y, (uint32_t)cw);
      yx = (uint32_t)((y << 5) | x);
      pass = fast_compass_px(I + y * kWImgPitch + x, kWImgPitch, min_th);
    }
    const uint64_t m = __ballot(pass);
    if (pass) L.queue[nq + __popcll(m & lt_mask)] = (uint16_t)yx;
    nq += __popcll(m);
  }
  __builtin_amdgcn_wave_barrier();
  // arc score of the survivors
  for (int base = 0; base < nq; base += 64) {
    const int i = base + lane;
    if (i < nq) {
      const int yx = L.queue[i], y = yx >> 5, x = yx & 31;
      const int s = fast_arc_score_px(I + y * kWImgPitch + x, kWImgPitch);
      S[y * kWSPitch + x] = (uint8_t)(s > min_th ? min(s, 255) : 0);
    }
  }
  __builtin_amdgcn_wave_barrier();
  // 8-neighbour suppression inside the cell (the border of the score plane is zero: pixels outside the cell do not compete)
  int n = 0;
  bool strong = false;
  for (int base = 0; base < nq; base += 64) {
    const int i = base + lane;
    bool ismax = false;
    uint32_t key = 0;
    int s = 0;
    if (i < nq) {
      const int yx = L.queue[i], y = yx >> 5, x = yx & 31;
      const uint8_t* sp = S + y * kWSPitch + x;
      s = sp[0];
      if (s != 0) {
        const int n0 = sp[-kWSPitch - 1], n1 = sp[-kWSPitch], n2 = sp[-kWSPitch + 1], n3 = sp[-1], n4 = sp[1],
                  n5 = sp[kWSPitch - 1], n6 = sp[kWSPitch], n7 = sp[kWSPitch + 1];
        ismax = max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7))) < s;
        key = ((uint32_t)s << 24) | ((uint32_t)(y0 + y) << 12) | (uint32_t)(x0 + x);
      }
    }
    const uint64_t m = __ballot(ismax);
    if (ismax) list[n + __popcll(m & lt_mask)] = key;
    n += __popcll(m);
    strong = strong || __ballot(ismax && s > ini_th) != 0ull;
  }
  __builtin_amdgcn_wave_barrier();
  // a strong corner silences the weak ones of the cell
  int kept = 0;
  for (int base = 0; base < n; base += 64) {
    const int e = base + lane;
    kept += __popcll(__ballot(e < n && (!strong || (int)(list[e] >> 24) > ini_th)));
  }
  if (kept == 0) return;
  uint32_t slot0 = 0;
  if (lane == 0) slot0 = atomicAdd(&key_cnt[b * kMaxL + level], (uint32_t)kept);
  slot0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot0);
  uint32_t* out = keys + (size_t)b * keys_per_frame;
  for (int base = 0; base < n; base += 64) {
    const int e = base + lane;
    const uint32_t v = e < n ? list[e] : 0u;
    const bool keep = e < n && (!strong || (int)(v >> 24) > ini_th);
    const uint64_t m = __ballot(keep);
    if (keep) {
      const uint32_t slot = slot0 + (uint32_t)__popcll(m & lt_mask);
      if (slot < cap) out[slot] = v;
      else atomicOr(flags, 1u);  // reported by gh_qt_check: never a silent drop
    }
    slot0 += (uint32_t)__popcll(m);
  }
}

// step 4' from a SCORE PLANE (round 5): orb.hip's 64 x 64 tile kernel -- the default mode's FAST passes: SWAR compass test,
// packed arc scores, the next pyramid level fused -- writes S (oracle step 2 / 3: the score where it exceeds min_th, else 0)
// for the whole level (fast_cells_kernel<.., PLANE>), and a wave per cell only does what the cell decides: 8-neighbour
// suppression against a zero border, the 20 -> 7 fallback, the key list.  The oracle is written the same way
// (oracle_orb_slam_candidates reads S).  Against slam_cells_wave_kernel the image is read once per level instead of 1.44
// times (36 x 36 tiles for 30 x 30 cells), the compass test runs 4 pixels per instruction, and seven resize launches go.
// Plane pixel (y, x) of a frame lives at plane[y * pitch + x + kQtPlaneX] (orb_quadtree.h).
// Two size classes (template BIG): cells up to 32 x 32 -- every level of a large image -- and up to 40 x 40: ORB-SLAM's cell is
// ceil(W / floor(W / 30)) pixels, i.e. 33 .. 36 where a side holds fewer than 15 cells (the upper levels of 1080p, most levels of
// VGA); until round 5b those levels fell back to the image-based one-workgroup-per-cell kernel (0.11 of 2.2 ms at 1080p).
constexpr int kPSPitch = 48;  // LDS score rows: a zero dword, up to nine (BIG: eleven) data dwords, slack
constexpr int kPlaneCellsPerWave = 4;  // cells a wave takes one after the other (see the slot reservation below)
constexpr int kPlaneOut = 128;         // kept keys a wave buffers before it must reserve slots by itself
constexpr int kPcBig = 40;
template <bool BIG>
struct PlaneCellLds {
  static constexpr int kPc = BIG ? kPcBig : kWcMax;
  __attribute__((aligned(16))) uint32_t S[(kPc + 2) * (kPSPitch / 4)];  // rows -1 .. kPc of the cell, zero outside it: 1632 B
  uint16_t queue[kPc * kPc];            // the cell's scored pixels, (y << 5) | x  (BIG: y << 6)
  uint32_t list[(kPc / 2) * (kPc / 2)];  // its suppressed maxima (pairwise non-adjacent: at most 16 x 16 / 20 x 20)
  uint32_t obuf[kPlaneOut];             // kept keys of the wave's cells until the workgroup reserves their slots
};
static_assert(sizeof(PlaneCellLds<false>) * 4 <= 160 * 1024 / 7, "seven workgroups per CU");
static_assert(sizeof(PlaneCellLds<false>::S) % 16 == 0 && sizeof(PlaneCellLds<true>::S) % 16 == 0, "cleared by 16-byte stores");
static_assert(4 + 3 + kPcBig + 1 <= kPSPitch, "zero dword + alignment + the widest cell fit a score row");

// SLOT RESERVATION.  Every cell of a (frame, level) appends to one key list through one counter; with a returning atomic per
// cell -- 2108 cells of a 1080p level 0 on ONE address -- the atomics were two thirds of the kernel (1.54 ms of which 1.04 ms,
// measured by replacing the reservation with fixed slots).  So a workgroup takes 16 cells (4 waves x 4 cells, one after the
// other), every wave buffers what its cells keep, and ONE atomic per workgroup reserves the slots of all of them (the order of
// the keys is irrelevant: every later step is a function of the key set).  A wave whose buffer fills up (> 128 kept keys in
// four cells: dense noise) reserves for itself and goes on.
template <bool BIG>
__global__ __launch_bounds__(256, BIG ? 5 : 7) void slam_cells_plane_kernel(LevelView pl, int ncols, int ncells, int wc, int hc, int ini_th,
                                                               uint32_t* __restrict__ keys, size_t keys_per_frame, uint32_t cap,
                                                               uint32_t* __restrict__ key_cnt, int level,
                                                               uint32_t* __restrict__ flags) {
  constexpr int kSlots = BIG ? 11 : 9;                              // data dwords per row
  constexpr int kLoadTrips = BIG ? (kPcBig * 11 + 63) / 64 : 5;     // 64-lane trips over rows x kSlots
  constexpr int kQShift = BIG ? 6 : 5;
  __shared__ PlaneCellLds<BIG> sh[4];
  __shared__ uint32_t wave_tot[4], wg_base;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, b = blockIdx.y;
  PlaneCellLds<BIG>& L = sh[wv];
  const uint8_t* plane = pl.base + (size_t)b * pl.frame_stride;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  uint32_t* out = keys + (size_t)b * keys_per_frame;
  uint32_t* list = L.list;
  int nout = 0;  // keys in L.obuf (wave-uniform)
  auto put_out = [&](uint32_t base, int count) {  // obuf[0 .. count) -> out[base ..]
    for (int e = lane; e < count; e += 64) {
      const uint32_t slot = base + (uint32_t)e;
      if (slot < cap) out[slot] = L.obuf[e];
      else atomicOr(flags, 1u);  // reported by gh_qt_check: never a silent drop
    }
  };
  for (int cc = 0; cc < kPlaneCellsPerWave; ++cc) {
    const int cell = ((int)blockIdx.x * 4 + wv) * kPlaneCellsPerWave + cc;
    if (cell >= ncells) break;  // (wave-uniform)
    const int ci = cell / ncols, cj = cell - ci * ncols;
    const int x0 = kEdge + cj * wc, y0 = kEdge + ci * hc;
    const int x1 = min(x0 + wc, pl.w - kEdge), y1 = min(y0 + hc, pl.h - kEdge);
    const int cw = x1 - x0, ch = y1 - y0;
    if (cw <= 0 || ch <= 0) continue;
    // rows y0 .. y1 - 1 as the aligned dwords that cover plane bytes of columns x0 .. x1 - 1; bytes of neighbouring cells masked to zero.
    // All (at most five) loads of a lane are asked for at once, the LDS plane is cleared under them.
    const int xa = (x0 + kQtPlaneX) & ~3, al = (x0 + kQtPlaneX) & 3;
    const int ndw = (al + cw + 3) >> 2;  // <= kSlots
    uint32_t vv[kLoadTrips];
    int slot[kLoadTrips];
#pragma unroll
    for (int t = 0; t < kLoadTrips; ++t) {
      const int idx = lane + 64 * t;
      // idx / 9: exact for idx < 320 (7282 / 65536 - 1 / 9 = 1.7e-6); idx / 11: exact for idx < 448 (5958 / 65536 - 1 / 11 = 2.8e-6)
      const int r = (int)(__umul24((uint32_t)idx, BIG ? 5958u : 7282u) >> 16), d = idx - r * kSlots;
      const bool on = idx < ch * kSlots && d < ndw;
      const int rc = on ? r : 0, dc = on ? d : 0;  // (clamped: a load without a branch around it)
      const uint32_t v = *reinterpret_cast<const uint32_t*>(plane + (__umul24((uint32_t)(y0 + rc), (uint32_t)pl.pitch) + (uint32_t)(xa + 4 * dc)));
      const int lo = al - 4 * dc, hi = al + cw - 4 * dc;  // keep bytes lo <= k < hi
      uint32_t m = 0xFFFFFFFFu;
      if (lo > 0) m &= lo >= 4 ? 0u : (0xFFFFFFFFu << (8 * lo));
      if (hi < 4) m &= hi <= 0 ? 0u : (0xFFFFFFFFu >> (8 * (4 - hi)));
      vv[t] = v & m;
      slot[t] = on ? (rc + 1) * (kPSPitch / 4) + 1 + dc : -1;
    }
    {
      uint4* s4 = reinterpret_cast<uint4*>(L.S);
      for (int idx = lane; idx < (int)(sizeof(L.S) / 16); idx += 64) s4[idx] = uint4{0u, 0u, 0u, 0u};
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < kLoadTrips; ++t)
      if (slot[t] >= 0) L.S[slot[t]] = vv[t];
    __builtin_amdgcn_wave_barrier();
    uint8_t* S = reinterpret_cast<uint8_t*>(L.S) + kPSPitch + 4 + al;  // score of cell pixel (0, 0)
    // the scored pixels: a unit = 16 bytes of a row (small cells: lane = (row, half); BIG: three parts per row, two trips of 64
    // units); positions appended by a prefix sum over the units' counts (the order of the queue is irrelevant)
    int nq = 0;
#pragma unroll
    for (int t = 0; t < (BIG ? 2 : 1); ++t) {
      const int u = lane + 64 * t;
      const int r = BIG ? (int)(__umul24((uint32_t)u, 43u) >> 7) : u >> 1;  // u / 3: exact for u < 128
      const int c0 = 16 * (BIG ? u - 3 * r : u & 1);
      uint32_t nzb = 0;
      if (r < ch && c0 < cw) {
        const uint32_t* q = L.S + (r + 1) * (kPSPitch / 4) + 1 + ((al + c0) >> 2);
        const uint32_t shb = (uint32_t)(al & 3);  // (c0 is a multiple of 4: the byte offset inside the dword is al)
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
        const uint32_t w4[4] = {__builtin_amdgcn_alignbyte(d1, d0, shb), __builtin_amdgcn_alignbyte(d2, d1, shb),
                                __builtin_amdgcn_alignbyte(d3, d2, shb), __builtin_amdgcn_alignbyte(d4, d3, shb)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t w = w4[j];
          const uint32_t f = ((((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) >> 7) & 0x01010101u;  // byte k -> 1 if non-zero
          nzb |= __builtin_amdgcn_udot4(f, 0x08040201u, 0u, false) << (4 * j);
        }
        // (BIG: the third part's dwords run past the row into the next one -- only columns < cw count)
        if constexpr (BIG) nzb &= cw - c0 >= 16 ? 0xFFFFu : ((1u << (cw - c0)) - 1u);
      }
      const int cnt = __popc(nzb);
      int incl = cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int tt = __shfl_up(incl, o);
        if (lane >= o) incl += tt;
      }
      int pos = nq + incl - cnt;
      nq += __shfl(incl, 63);
      const uint32_t rc0 = (uint32_t)((r << kQShift) | c0);
      while (nzb) {
        const int k = __ffs((int)nzb) - 1;
        L.queue[pos++] = (uint16_t)(rc0 + (uint32_t)k);
        nzb &= nzb - 1u;
      }
    }
    __builtin_amdgcn_wave_barrier();
    // 8-neighbour suppression inside the cell (the border of the score plane is zero: pixels outside the cell do not compete)
    int n = 0;
    bool strong = false;
    for (int base = 0; base < nq; base += 64) {
      const int i = base + lane;
      bool ismax = false;
      uint32_t key = 0;
      int sc = 0;
      if (i < nq) {
        const int yx = L.queue[i], y = yx >> kQShift, x = yx & ((1 << kQShift) - 1);
        const uint8_t* sp = S + y * kPSPitch + x;
        sc = sp[0];
        const int n0 = sp[-kPSPitch - 1], n1 = sp[-kPSPitch], n2 = sp[-kPSPitch + 1], n3 = sp[-1], n4 = sp[1],
                  n5 = sp[kPSPitch - 1], n6 = sp[kPSPitch], n7 = sp[kPSPitch + 1];
        ismax = max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7))) < sc;
        key = ((uint32_t)sc << 24) | ((uint32_t)(y0 + y) << 12) | (uint32_t)(x0 + x);
      }
      const uint64_t m = __ballot(ismax);
      if (ismax) list[n + __popcll(m & lt_mask)] = key;
      n += __popcll(m);
      strong = strong || __ballot(ismax && sc > ini_th) != 0ull;
    }
    __builtin_amdgcn_wave_barrier();
    // a strong corner silences the weak ones of the cell; what is kept goes to the wave's buffer
    for (int base = 0; base < n; base += 64) {
      const int e = base + lane;
      const uint32_t v = e < n ? list[e] : 0u;
      const bool keep = e < n && (!strong || (int)(v >> 24) > ini_th);
      const uint64_t m = __ballot(keep);
      const int kc = __popcll(m);
      if (nout + kc > kPlaneOut) {  // (wave-uniform) the buffer is full: this wave reserves the slots of what it holds
        uint32_t base0 = 0;
        if (lane == 0) base0 = atomicAdd(&key_cnt[b * kMaxL + level], (uint32_t)nout);
        base0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)base0);
        __builtin_amdgcn_wave_barrier();
        put_out(base0, nout);
        __builtin_amdgcn_wave_barrier();
        nout = 0;
      }
      if (keep) L.obuf[nout + __popcll(m & lt_mask)] = v;
      nout += kc;
    }
    __builtin_amdgcn_wave_barrier();
  }
  // one reservation for the workgroup's (up to) 16 cells
  if (lane == 0) wave_tot[wv] = (uint32_t)nout;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t tot = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    wg_base = tot ? atomicAdd(&key_cnt[b * kMaxL + level], tot) : 0u;
  }
  __syncthreads();
  uint32_t base0 = wg_base;
  for (int k = 0; k < wv; ++k) base0 += wave_tot[k];
  put_out(base0, nout);
}

// step 4', any cell size: one workgroup per cell
__global__ __launch_bounds__(256) void slam_cells_kernel(LevelView lv, int ncols, int wc, int hc, int min_th, int ini_th,
                                                         uint32_t* __restrict__ keys, size_t keys_per_frame,
                                                         uint32_t cap, uint32_t* __restrict__ key_cnt, int level,
                                                         uint32_t* __restrict__ flags) {
  __shared__ uint8_t s_img[(kCellMax + 6) * kCellPitch];
  __shared__ uint8_t s_S[kCellMax * 60];
  __shared__ uint32_t s_list[kCellList];
  __shared__ int s_n, s_strong, s_kept, s_base;
  const int tid = threadIdx.x, b = blockIdx.y;
  const int ci = (int)blockIdx.x / ncols, cj = (int)blockIdx.x - ci * ncols;
  const int x0 = kEdge + cj * wc, y0 = kEdge + ci * hc;
  const int x1 = min(x0 + wc, lv.w - kEdge), y1 = min(y0 + hc, lv.h - kEdge);
  const int cw = x1 - x0, ch = y1 - y0;
  if (cw <= 0 || ch <= 0) return;
  const uint8_t* img = lv.base + (size_t)b * lv.frame_stride;
  const int tw = cw + 6;
  for (int idx = tid; idx < (ch + 6) * tw; idx += 256) {
    const int r = idx / tw, c = idx - r * tw;
    s_img[r * kCellPitch + c] = img[(size_t)(y0 - 3 + r) * lv.pitch + (x0 - 3 + c)];
  }
  if (tid == 0) {
    s_n = 0;
    s_strong = 0;
    s_kept = 0;
  }
  __syncthreads();
  for (int p = tid; p < cw * ch; p += 256) {
    const int y = p / cw, x = p - y * cw;
    const int s = fast_score_px(&s_img[(y + 3) * kCellPitch + x + 3], kCellPitch, min_th);
    s_S[y * 60 + x] = (uint8_t)(s > min_th ? min(s, 255) : 0);
  }
  __syncthreads();
  for (int p = tid; p < cw * ch; p += 256) {
    const int y = p / cw, x = p - y * cw;
    const int s = s_S[y * 60 + x];
    if (s == 0) continue;
    bool ismax = true;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int xx = x + dx, yy = y + dy;
        if ((dx == 0 && dy == 0) || xx < 0 || xx >= cw || yy < 0 || yy >= ch) continue;
        if ((int)s_S[yy * 60 + xx] >= s) ismax = false;
      }
    if (!ismax) continue;
    s_list[atomicAdd(&s_n, 1)] = ((uint32_t)s << 24) | ((uint32_t)(y0 + y) << 12) | (uint32_t)(x0 + x);
    if (s > ini_th) s_strong = 1;
  }
  __syncthreads();
  const int n = s_n;
  const bool strong = s_strong != 0;
  for (int e = tid; e < n; e += 256)
    if (!strong || (int)(s_list[e] >> 24) > ini_th) atomicAdd(&s_kept, 1);
  __syncthreads();
  if (s_kept == 0) return;
  if (tid == 0) s_base = (int)atomicAdd(&key_cnt[b * kMaxL + level], (uint32_t)s_kept);
  __syncthreads();
  uint32_t* out = keys + (size_t)b * keys_per_frame;
  for (int e = tid; e < n; e += 256) {
    const uint32_t v = s_list[e];
    if (strong && (int)(v >> 24) <= ini_th) continue;
    const uint32_t slot = (uint32_t)s_base + (uint32_t)atomicAdd(&s_n, 1) - (uint32_t)n;  // (s_n keeps counting past n)
    if (slot < cap) out[slot] = v;
    else atomicOr(flags, 1u);  // reported by gh_qt_check: never a silent drop
  }
}

// ---- step 5'
// exclusive scan of one int per thread over the workgroup of THREADS lanes
template <int THREADS>
__device__ __forceinline__ int block_scan_wg(int v, int* wave_tot /* shared[16] */, int* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  __syncthreads();
  if (lane == 63) wave_tot[wv] = incl;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < THREADS / 64; ++k) {
    const int t = wave_tot[k];
    if (k < wv) off += t;
    tot += t;
  }
  *total = tot;
  return off + incl - v;
}

// ascending bitonic sort of NODES values in LDS (every thread of the NODES / 2 of the workgroup calls it)
template <int NODES, typename T>
__device__ __forceinline__ void bitonic_sort_nodes(T* a) {
  for (int k = 2; k <= NODES; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      const int t = threadIdx.x;                     // pair index: NODES / 2 pairs
      const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // lower element of the pair
      const int p = i | j;
      const bool up = (i & k) == 0;
      const T x = a[i], y = a[p];
      if ((x > y) == up) {
        a[i] = y;
        a[p] = x;
      }
    }
  __syncthreads();
}

// (NODES = table capacity: quota + 3 and 4 * roots must fit; the workgroup has NODES / 2 threads, each owns two table rows)
template <int NODES>
struct QtShared {
  uint32_t box_x[2][NODES];   // x0 | x1 << 16 (region coordinates)
  uint32_t box_y[2][NODES];
  uint32_t cnt[2][NODES];     // keys of the node
  uint8_t fresh[2][NODES];    // created by the last pass
  uint32_t ccnt[NODES][4];    // keys of the four children of a node being split
  uint16_t cbase[NODES];      // id of the node (or of its first child) in the next table
  uint8_t split[NODES];
  unsigned long long sortbuf[NODES];
  int wave_tot[16];
  int cut;
};
static_assert(sizeof(QtShared<2048>) <= 120 * 1024 && sizeof(QtShared<1024>) <= 160 * 1024 / 3 && sizeof(QtShared<512>) <= 160 * 1024 / 5,
              "workgroups per CU by LDS: 1 / 3 / 5 (by threads: 2 / 4 / 8)");

__device__ __forceinline__ int qt_quadrant(uint32_t bx, uint32_t by, uint32_t key) {
  const int x0 = (int)(bx & 0xFFFFu), x1 = (int)(bx >> 16), y0 = (int)(by & 0xFFFFu), y1 = (int)(by >> 16);
  const int xm = x0 + ((x1 - x0 + 1) >> 1), ym = y0 + ((y1 - y0 + 1) >> 1);
  const int kx = (int)(key & 0xFFFu) - kBorder, ky = (int)((key >> 12) & 0xFFFu) - kBorder;
  return (kx < xm ? 0 : 1) + (ky < ym ? 0 : 2);
}

template <int NODES>
__global__ __launch_bounds__(NODES / 2) void slam_quadtree_kernel(QtArgs a, const uint32_t* __restrict__ keys,
                                                                   uint16_t* __restrict__ knode, size_t keys_per_frame,
                                                                   const uint32_t* __restrict__ key_cnt, int K,
                                                                   SelKp* __restrict__ sel, int32_t* __restrict__ level_cnt) {
  extern __shared__ __attribute__((aligned(16))) uint8_t qt_lds[];
  constexpr int kQtNodes = NODES, kQtThreads = NODES / 2;
  QtShared<NODES>& sh = *reinterpret_cast<QtShared<NODES>*>(qt_lds);
  const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const QtLevel L = a.lv[l];
  const int N = L.quota;
  const uint32_t m = min(key_cnt[b * kMaxL + l], L.cap);
  if (N <= 0 || m == 0 || L.cap == 0) {
    if (tid == 0) level_cnt[b * kMaxL + l] = 0;
    return;
  }
  const uint32_t* kk = keys + (size_t)b * keys_per_frame + L.key_off;
  uint16_t* kn = knode + (size_t)b * keys_per_frame + L.key_off;
  const int H2 = L.h - 2 * kBorder;
  int cur = 0, n_tab = L.n_ini;
  // roots
  for (int n = tid; n < kQtNodes; n += kQtThreads) {
    if (n < L.n_ini) {
      const int x0 = (int)__fmul_rn(L.hx, (float)n), x1 = (int)__fmul_rn(L.hx, (float)(n + 1));
      sh.box_x[0][n] = (uint32_t)x0 | ((uint32_t)x1 << 16);
      sh.box_y[0][n] = (uint32_t)H2 << 16;
    }
    sh.cnt[0][n] = 0;
    sh.fresh[0][n] = 0;
  }
  __syncthreads();
  for (uint32_t k = tid; k < m; k += kQtThreads) {
    const int kx = (int)(kk[k] & 0xFFFu) - kBorder;
    int r = (int)__fdiv_rn((float)kx, L.hx);
    r = min(r, L.n_ini - 1);
    kn[k] = (uint16_t)r;
    atomicAdd(&sh.cnt[0][r], 1u);
  }
  __syncthreads();
  int n_nodes = 0;
  {
    int mine = 0;
    for (int n = tid; n < n_tab; n += kQtThreads) mine += sh.cnt[0][n] > 0;
    block_scan_wg<NODES / 2>(mine, sh.wave_tot, &n_nodes);
  }
  // one pass: the children of every node with split[] set have been counted into ccnt; builds the next table and moves the keys.
  // Returns the new node count; *made = children holding more than one key.
  auto rebuild = [&](int* made) -> int {
    const int nx = cur ^ 1;
    // thread t owns table rows 2 t and 2 t + 1 (table order = scan order)
    int c[2] = {0, 0}, e = 0;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int n = 2 * tid + u;
      if (n < n_tab) {
        if (sh.split[n]) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            c[u] += sh.ccnt[n][q] > 0;
            e += sh.ccnt[n][q] > 1;
          }
        } else {
          c[u] = sh.cnt[cur][n] > 0;
        }
      }
    }
    int total, made_total;
    const int base = block_scan_wg<NODES / 2>(c[0] + c[1], sh.wave_tot, &total);
    block_scan_wg<NODES / 2>(e, sh.wave_tot, &made_total);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int n = 2 * tid + u;
      if (n >= n_tab) continue;
      int id = base + (u ? c[0] : 0);
      sh.cbase[n] = (uint16_t)id;
      const uint32_t bx = sh.box_x[cur][n], by = sh.box_y[cur][n];
      if (sh.split[n]) {
        const uint32_t x0 = bx & 0xFFFFu, x1 = bx >> 16, y0 = by & 0xFFFFu, y1 = by >> 16;
        const uint32_t xm = x0 + ((x1 - x0 + 1) >> 1), ym = y0 + ((y1 - y0 + 1) >> 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t cc = sh.ccnt[n][q];
          if (cc == 0) continue;
          sh.box_x[nx][id] = (q & 1) ? (xm | (x1 << 16)) : (x0 | (xm << 16));
          sh.box_y[nx][id] = (q & 2) ? (ym | (y1 << 16)) : (y0 | (ym << 16));
          sh.cnt[nx][id] = cc;
          sh.fresh[nx][id] = 1;
          ++id;
        }
      } else if (sh.cnt[cur][n] > 0) {
        sh.box_x[nx][id] = bx;
        sh.box_y[nx][id] = by;
        sh.cnt[nx][id] = sh.cnt[cur][n];
        sh.fresh[nx][id] = 0;
      }
    }
    __syncthreads();
    for (uint32_t k = tid; k < m; k += kQtThreads) {
      const int n = kn[k];
      int id = sh.cbase[n];
      if (sh.split[n]) {
        const int q = qt_quadrant(sh.box_x[cur][n], sh.box_y[cur][n], kk[k]);
        for (int qq = 0; qq < q; ++qq) id += sh.ccnt[n][qq] > 0;
      }
      kn[k] = (uint16_t)id;
    }
    __syncthreads();
    cur = nx;
    n_tab = total;
    *made = made_total;
    return total;
  };
  // counts the children of the nodes with split[] set
  auto count_children = [&]() {
    for (int n = tid; n < n_tab; n += kQtThreads)
      if (sh.split[n]) sh.ccnt[n][0] = sh.ccnt[n][1] = sh.ccnt[n][2] = sh.ccnt[n][3] = 0;
    __syncthreads();
    for (uint32_t k = tid; k < m; k += kQtThreads) {
      const int n = kn[k];
      if (sh.split[n]) atomicAdd(&sh.ccnt[n][qt_quadrant(sh.box_x[cur][n], sh.box_y[cur][n], kk[k])], 1u);
    }
    __syncthreads();
  };

  bool finish = false;
  while (!finish) {
    // (A) every node that holds more than one key is split
    const int before = n_nodes;
    for (int n = tid; n < n_tab; n += kQtThreads) sh.split[n] = sh.cnt[cur][n] > 1;
    __syncthreads();
    count_children();
    int made;
    n_nodes = rebuild(&made);
    if (n_nodes >= N || n_nodes == before) {
      finish = true;
    } else if (n_nodes + 3 * made > N) {
      // (B) the nodes of the last pass, largest first, one by one until N is reached
      while (!finish) {
        const int before_b = n_nodes;
        for (int n = tid; n < kQtNodes; n += kQtThreads) {
          const bool cand = n < n_tab && sh.fresh[cur][n] && sh.cnt[cur][n] > 1;
          if (n < n_tab) sh.split[n] = cand;
          const uint32_t bx = sh.box_x[cur][n < n_tab ? n : 0], by = sh.box_y[cur][n < n_tab ? n : 0];
          sh.sortbuf[n] = cand ? ((unsigned long long)(0x3FFFFFu - sh.cnt[cur][n]) << 35) | ((unsigned long long)(by & 0xFFFu) << 23) |
                                     ((unsigned long long)(bx & 0xFFFu) << 11) | (unsigned long long)n
                               : ~0ull;
        }
        if (tid == 0) sh.cut = kQtNodes;
        __syncthreads();
        count_children();
        bitonic_sort_nodes<NODES>(sh.sortbuf);
        // gains in expansion order; the first position where the node count reaches N ends the pass
        int g[2] = {0, 0};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const unsigned long long v = sh.sortbuf[2 * tid + u];
          if (v != ~0ull) {
            const int n = (int)(v & 0x7FFu);
            g[u] = (sh.ccnt[n][0] > 0) + (sh.ccnt[n][1] > 0) + (sh.ccnt[n][2] > 0) + (sh.ccnt[n][3] > 0) - 1;
          }
        }
        int tot;
        const int excl = block_scan_wg<NODES / 2>(g[0] + g[1], sh.wave_tot, &tot);
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (sh.sortbuf[2 * tid + u] != ~0ull && n_nodes + excl + g[0] + (u ? g[1] : 0) >= N) atomicMin(&sh.cut, 2 * tid + u);
        __syncthreads();
        const int cut = sh.cut;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const unsigned long long v = sh.sortbuf[2 * tid + u];
          if (v != ~0ull && 2 * tid + u > cut) sh.split[(int)(v & 0x7FFu)] = 0;
        }
        __syncthreads();
        n_nodes = rebuild(&made);
        if (n_nodes >= N || n_nodes == before_b) finish = true;
      }
    }
  }
  // the best key of every node: (S desc, y asc, x asc) = max of S | ~(y, x)
  uint32_t* best = &sh.ccnt[0][0];
  for (int n = tid; n < kQtNodes; n += kQtThreads) best[n] = 0;
  __syncthreads();
  for (uint32_t k = tid; k < m; k += kQtThreads) {
    const uint32_t v = kk[k];
    atomicMax(&best[kn[k]], (v & 0xFF000000u) | (0xFFFFFFu - (v & 0xFFFFFFu)));
  }
  __syncthreads();
  // the tree may hold up to 3 nodes more than N: keep the N best, then (y, x) order
  uint32_t* srt = reinterpret_cast<uint32_t*>(sh.sortbuf);
  for (int n = tid; n < kQtNodes; n += kQtThreads) srt[n] = ~best[n];  // ascending ~ = descending (S, ~yx); empty slots last
  bitonic_sort_nodes<NODES>(srt);
  const int keep = min(n_tab, N);
  {
    uint32_t v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = 2 * tid + u;
      const uint32_t w = ~srt[i];  // S << 24 | ~yx
      v[u] = i < keep ? ((0xFFFFFFu - (w & 0xFFFFFFu)) << 8) | (w >> 24) : 0xFFFFFFFFu;
    }
    __syncthreads();
    srt[2 * tid] = v[0];
    srt[2 * tid + 1] = v[1];
  }
  bitonic_sort_nodes<NODES>(srt);
  SelKp* out = sel + (size_t)b * K + L.quota_off;
  for (int i = tid; i < keep; i += kQtThreads) {
    const uint32_t v = srt[i];
    SelKp o;
    o.x = (uint16_t)((v >> 8) & 0xFFFu);
    o.y = (uint16_t)(v >> 20);
    o.score = (uint8_t)(v & 0xFFu);
    o.level = (uint8_t)l;
    o.pad = 0;
    out[i] = o;
  }
  if (tid == 0) level_cnt[b * kMaxL + l] = keep;
}

}  // namespace

struct gh_qt_plan {
  int L = 0, max_batch = 0;
  QtArgs args{};
  size_t keys_per_frame = 0;
  uint32_t* keys = nullptr;
  uint16_t* knode = nullptr;
  uint32_t* key_cnt = nullptr;  // [max_batch][8], then the overflow flag word
  bool attr_set = false;
  int nodes = 2048;  // least table capacity of the tree kernel (512 / 1024 / 2048) that holds every level's quota + 3 and 4 roots
  int nodes_forced = 0;  // GSLAM_HIP_QT_NODES
  bool can_overflow = true;  // the key lists are smaller than the worst case (the plan's key budget cut them): calls must check the flag
};

void gh_qt_destroy(gh_qt_plan* q) {
  if (!q) return;
  if (q->keys) (void)hipFree(q->keys);
  if (q->knode) (void)hipFree(q->knode);
  if (q->key_cnt) (void)hipFree(q->key_cnt);
  delete q;
}

gh_status gh_qt_create(gh_ctx* ctx, int n_levels, const int* lw, const int* lh, const int* quota, int max_batch,
                       gh_qt_plan** out, size_t* bytes) {
  *out = nullptr;
  for (int i = 0; i < 16; ++i)
    if (kRing.dx[i] != GH_ORB_RING[i][0] || kRing.dy[i] != GH_ORB_RING[i][1])
      return gh_set_error(ctx, GH_ERR_UNSUPPORTED, "orb_quadtree.hip: ring table differs from gslam_orb_tables.h");
  gh_qt_plan* q = new (std::nothrow) gh_qt_plan();
  if (!q) return GH_ERR_NOMEM;
  q->L = n_levels;
  q->max_batch = max_batch;
  // key slots: the exact bound (suppressed maxima of a cell are pairwise non-adjacent) while it fits the budget, a share
  // of the budget otherwise -- an overflowing list is an error of the call (gh_qt_check), never a silent truncation
  size_t worst[kMaxL] = {}, worst_total = 0;
  int need_nodes = 0;
  bool any_cut = false;
  int qo = 0;
  for (int l = 0; l < kMaxL; ++l) {
    QtLevel& v = q->args.lv[l];
    v = QtLevel{};
    if (l >= n_levels) continue;
    v.w = lw[l];
    v.h = lh[l];
    v.quota = quota[l];
    v.quota_off = qo;
    qo += quota[l];
    if (v.w <= 2 * kEdge || v.h <= 2 * kEdge || v.quota <= 0) {
      v.quota = 0;
      continue;
    }
    if (v.w > 4096 || v.h > 4096 || v.quota + 3 > kQtNodes) {
      gh_qt_destroy(q);
      return gh_set_error(ctx, GH_ERR_ARG,
                          "quadtree distribution: level %d is %d x %d with quota %d (limits: 4096 x 4096, quota <= %d)", l,
                          v.w, v.h, v.quota, kQtNodes - 3);
    }
    const int W2 = v.w - 2 * kBorder, H2 = v.h - 2 * kBorder;
    v.ncols = W2 / 30 > 1 ? W2 / 30 : 1;
    v.nrows = H2 / 30 > 1 ? H2 / 30 : 1;
    v.wc = (W2 + v.ncols - 1) / v.ncols;
    v.hc = (H2 + v.nrows - 1) / v.nrows;
    v.n_ini = (int)roundf((float)W2 / (float)H2);
    if (v.n_ini < 1) v.n_ini = 1;
    v.hx = (float)W2 / (float)v.n_ini;
    if (4 * v.n_ini > kQtNodes || v.wc > kCellMax || v.hc > kCellMax) {
      gh_qt_destroy(q);
      return gh_set_error(ctx, GH_ERR_ARG, "quadtree distribution: level %d (%d x %d) has an unsupported aspect ratio", l, v.w, v.h);
    }
    worst[l] = (size_t)v.ncols * v.nrows * ((v.wc + 1) / 2) * ((v.hc + 1) / 2);
    worst_total += worst[l];
    need_nodes = need_nodes > v.quota + 3 ? need_nodes : v.quota + 3;
    need_nodes = need_nodes > 4 * v.n_ini ? need_nodes : 4 * v.n_ini;
  }
  // The tree kernel is one workgroup of nodes / 2 threads per (frame, level), bound by the latency of its passes: a smaller table
  // means more workgroups per CU at once (LDS 106 / 53 / 27 KB: 1 / 3 / 5 per CU; GSLAM_HIP_QT_NODES forces a larger one for A/B runs)
  q->nodes = need_nodes <= 512 ? 512 : (need_nodes <= 1024 ? 1024 : 2048);
  if (const char* e = getenv("GSLAM_HIP_QT_NODES")) {
    const int f = atoi(e);
    if ((f == 512 || f == 1024 || f == 2048) && f >= q->nodes) q->nodes_forced = f;
  }
  const size_t budget = ((size_t)4 << 30) / 6 / (size_t)max_batch;  // 4 GB for keys (4 B) + node ids (2 B) of all frames
  size_t off = 0;
  for (int l = 0; l < n_levels; ++l) {
    QtLevel& v = q->args.lv[l];
    size_t cap = worst[l];
    if (worst_total > budget) cap = (size_t)((double)worst[l] * (double)budget / (double)worst_total);
    if (cap >= ((size_t)1 << 22)) cap = ((size_t)1 << 22) - 1;  // the sort key of stage (B) carries the count in 22 bits
    if (cap < worst[l]) any_cut = true;
    v.cap = (uint32_t)cap;
    v.key_off = off;
    off += (cap + 63) & ~(size_t)63;
  }
  q->keys_per_frame = off ? off : 64;
  q->can_overflow = any_cut;
  const size_t B = (size_t)max_batch;
  gh_status st;
  if ((st = gh_dev_alloc(ctx, B * q->keys_per_frame * 4, (void**)&q->keys)) != GH_OK ||
      (st = gh_dev_alloc(ctx, B * q->keys_per_frame * 2, (void**)&q->knode)) != GH_OK ||
      (st = gh_dev_alloc(ctx, (B * kMaxL + 1) * 4, (void**)&q->key_cnt)) != GH_OK) {
    gh_qt_destroy(q);
    return st;
  }
  if (hipMemset(q->key_cnt, 0, (B * kMaxL + 1) * 4) != hipSuccess) {
    gh_qt_destroy(q);
    return gh_set_error(ctx, GH_ERR_HIP, "gh_qt_create: hipMemset failed");
  }
  if (bytes) *bytes += B * q->keys_per_frame * 6 + (B * kMaxL + 1) * 4;
  *out = q;
  return GH_OK;
}

// The three parts of gh_qt_enqueue, for a caller that wants the cells of level l on another stream than the kernel that
// produces level l + 1 (orb.hip): begin (counters), cells of one level, tree.  All on ctx->stream at the time of the call.
gh_status gh_qt_begin(gh_ctx* ctx, gh_qt_plan* q, int batch) {
  GH_CHECK_ARG(ctx, q && batch >= 1 && batch <= q->max_batch);
  if (!q->attr_set) {
    GH_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(slam_quadtree_kernel<2048>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(QtShared<2048>)));
    q->attr_set = true;
  }
  GH_HIP(ctx, hipMemsetAsync(q->key_cnt, 0, ((size_t)batch * kMaxL) * 4, ctx->stream));
  return GH_OK;
}

gh_status gh_qt_cells(gh_ctx* ctx, gh_qt_plan* q, int l, const LevelView& img, const LevelView* plane, int batch, int min_th, int ini_th) {
  uint32_t* flags = q->key_cnt + (size_t)q->max_batch * kMaxL;
  const QtLevel& v = q->args.lv[l];
  if (v.quota <= 0) return GH_OK;
  static const bool wave_cells = getenv("GSLAM_HIP_QT_WAVE_CELLS") == nullptr || atoi(getenv("GSLAM_HIP_QT_WAVE_CELLS")) != 0;  // (A/B switch)
  if (plane != nullptr && plane->base != nullptr) {  // the level's score plane exists (gh_qt_plane_ok): cells from it
    if (v.wc <= kWcMax && v.hc <= kWcMax)
      GH_LAUNCH(ctx, "orb_slam_cells", slam_cells_plane_kernel<false>, dim3(gh_div_up(v.ncols * v.nrows, 4 * kPlaneCellsPerWave), batch),
                dim3(256), 0, *plane, v.ncols, v.ncols * v.nrows, v.wc, v.hc, ini_th, q->keys + v.key_off, q->keys_per_frame, v.cap,
                q->key_cnt, l, flags);
    else
      GH_LAUNCH(ctx, "orb_slam_cells", slam_cells_plane_kernel<true>, dim3(gh_div_up(v.ncols * v.nrows, 4 * kPlaneCellsPerWave), batch),
                dim3(256), 0, *plane, v.ncols, v.ncols * v.nrows, v.wc, v.hc, ini_th, q->keys + v.key_off, q->keys_per_frame, v.cap,
                q->key_cnt, l, flags);
  } else if (wave_cells && v.wc <= kWcMax && v.hc <= kWcMax && (img.pitch & 3) == 0 && (reinterpret_cast<uintptr_t>(img.base) & 3) == 0 &&
             (img.frame_stride & 3) == 0) {
    GH_LAUNCH(ctx, "orb_slam_cells", slam_cells_wave_kernel, dim3(gh_div_up(v.ncols * v.nrows, 4), batch), dim3(256), 0, img, v.ncols,
              v.ncols * v.nrows, v.wc, v.hc, min_th, ini_th, q->keys + v.key_off, q->keys_per_frame, v.cap, q->key_cnt, l, flags);
  } else {
    GH_LAUNCH(ctx, "orb_slam_cells", slam_cells_kernel, dim3(v.ncols * v.nrows, batch), dim3(256), 0, img, v.ncols, v.wc, v.hc, min_th,
              ini_th, q->keys + v.key_off, q->keys_per_frame, v.cap, q->key_cnt, l, flags);
  }
  return GH_OK;
}

gh_status gh_qt_tree(gh_ctx* ctx, gh_qt_plan* q, int batch, const int* quota_off, int K, SelKp* sel, int32_t* level_cnt) {
  for (int l = 0; l < q->L; ++l) GH_CHECK_ARG(ctx, q->args.lv[l].quota == 0 || q->args.lv[l].quota_off == quota_off[l]);
#define GH_QT_TREE(NODES_)                                                                                                     \
  GH_LAUNCH(ctx, "orb_slam_quadtree", slam_quadtree_kernel<NODES_>, dim3(q->L, batch), dim3(NODES_ / 2), sizeof(QtShared<NODES_>), \
            q->args, q->keys, q->knode, q->keys_per_frame, q->key_cnt, K, sel, level_cnt)
  // 256-thread workgroups win when there are many (VGA x 500 frames: 0.235 vs 0.39 ms), 512-thread ones when a launch is a few
  // hundred long trees (1080p x 100 frames: 0.30 vs 0.33 ms) -- profiles/orb_slam_mode_r05.txt
  const int nodes = q->nodes_forced ? q->nodes_forced : (q->nodes <= 512 && (long long)q->L * batch >= 2048 ? 512 : (q->nodes <= 1024 ? 1024 : 2048));
  if (nodes == 512) GH_QT_TREE(512);
  else if (nodes == 1024) GH_QT_TREE(1024);
  else GH_QT_TREE(2048);
#undef GH_QT_TREE
  return GH_OK;
}

gh_status gh_qt_enqueue(gh_ctx* ctx, gh_qt_plan* q, const LevelView* lv, int batch, int min_th, int ini_th,
                        const int* quota_off, int K, SelKp* sel, int32_t* level_cnt, const LevelView* planes) {
  GH_TRY(gh_qt_begin(ctx, q, batch));
  for (int l = 0; l < q->L; ++l) GH_TRY(gh_qt_cells(ctx, q, l, lv[l], planes ? &planes[l] : nullptr, batch, min_th, ini_th));
  return gh_qt_tree(ctx, q, batch, quota_off, K, sel, level_cnt);
}

gh_status gh_qt_check(gh_ctx* ctx, gh_qt_plan* q) {
  uint32_t f = 0;
  uint32_t* flags = q->key_cnt + (size_t)q->max_batch * kMaxL;
  GH_HIP(ctx, hipMemcpy(&f, flags, 4, hipMemcpyDeviceToHost));
  if (f != 0) {
    (void)hipMemset(flags, 0, 4);
    return gh_set_error(ctx, GH_ERR_NOMEM,
                        "quadtree distribution: a candidate list overflowed its buffer (more FAST maxima than the plan's "
                        "key budget holds for this batch size; use a smaller max_batch)");
  }
  return GH_OK;
}

// Whether a call has to wait for its kernels and read the overflow flag (gh_qt_check): only when the plan's key budget cut a
// level's list below its worst case -- ceil(wc / 2) ceil(hc / 2) strict maxima per cell, they are pairwise non-adjacent.
bool gh_qt_can_overflow(const gh_qt_plan* q) { return !q || q->can_overflow; }

// whether level l's cells can be taken from a score plane (cells of at most 40 x 40: a level whose side holds 3+ cells)
bool gh_qt_plane_ok(const gh_qt_plan* q, int l) {
  return q && l >= 0 && l < q->L && q->args.lv[l].quota > 0 && q->args.lv[l].wc <= kPcBig && q->args.lv[l].hc <= kPcBig;
}
