// Camera order of the reduced camera system (host code, once per graph topology).
//
// The band solver (chol_cr.hip) needs every point's observers within kBandSpan camera indices of each other, the arrowhead solver
// the same for all but a few "border" cameras.  Rounds 4-5 took the CALLER's camera order as given: a trajectory handed over in
// temporal order was a band, the same cameras in any other order went to the dense factorisation (C5: 188 -> 0.89 LM iterations
// per second).  But GSLAM::BundleGraph::keyframes is a plain vector whose vertices carry no id (GSLAM/core/Optimizer.h:116-119,
// 150-157) and a SLAM back end fills it in co-visibility order, not in temporal order; Ceres' SPARSE_SCHUR behind
// Optimizer::optimize (Optimizer.h:229) orders the reduced system for itself.  So does this:
//
//   1. the caller's order is a band already                      -> keep it (no permutation, nothing copied)
//   2. the caller's order is a band + a small border             -> arrow ordering of round 5 (far observers of long-range points last)
//   3. otherwise: BANDWIDTH-REDUCING ORDER of the camera co-visibility graph (covis_order below), then 1. / 2. on the new order
//
// Everything downstream is order-agnostic: perm[new] = old camera, the entry points of ba.hip translate poses / gauge masks /
// observation camera ids on the way in and poses on the way out (ArrowProblem).
//
// covis_order -- weighted maximum-adjacency order + barycentre refinement:
//   * adjacency from a SAMPLE of the points (every st-th point, at most ~2^20 observations: the order is a heuristic, the exact
//     span check that follows reads every observation); weight(a, b) = sampled points both cameras see.  Sampling doubles as a
//     weight filter: the handful of loop-closure points rarely make it into the sample, so the order follows the trajectory and
//     the closure points are dealt with by the arrow ordering afterwards, as in the caller-ordered case.
//   * start at a pseudo-peripheral camera (two breadth-first sweeps), then repeatedly number the unnumbered camera with the
//     largest total weight to the numbered ones (lazy max-heap).  On a trajectory (camera i shares points with i +- w) this walks
//     the trajectory in order from its end: the next camera along the line is always the one that shares most points with what
//     has been numbered.  Plain reverse Cuthill-McKee orders by breadth-first LEVEL and by degree inside a level, which leaves
//     neighbours up to two level widths apart -- 48 instead of 24 on the synthetic trajectories, beyond the solver's 31.
//   * up to eight barycentre sweeps (a camera moves to the mean position of its sampled points, a point sits at the mean position of
//     its observers; points whose observers are more than 4 * kBandSpan apart -- loop closures -- do not vote) iron out what
//     sampling noise scrambled locally.
#include "common.h"
#include "host_pool.h"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <queue>
#include <utility>
#include <vector>

constexpr int kBandSpan = 31;         // gh_cr_tiles: 6 * 31 + 5 = 191 <= 3 * 64
constexpr int kMaxBorderCams = 1024;  // 6144 border rows: beyond that the dense corner dominates
constexpr int kMaxBorderPts = 2048;   // the same 6144 rows when the border is made of POINTS (3 unknowns each)

namespace {

double now_ms() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// pool threads of the sampling pass and of the adjacency lists (GSLAM_HIP_BA_ORDER_THREADS; 1 = serial).  Measured on the GPU
// box's host (EPYC 9575F, C5 shuffled, 1 / 4 / 8 threads): sample lists 12.4 / 17.0 / 6.7 ms, adjacency 22.5 / 10.2 / 5.4 ms.
int order_threads() {
  static const int v = [] {
    const char* e = getenv("GSLAM_HIP_BA_ORDER_THREADS");
    const int t = e ? atoi(e) : 8;
    return t < 1 ? 1 : (t > 16 ? 16 : t);
  }();
  return v;
}
// ... and of the per-point camera range (GSLAM_HIP_BA_ORDER_RANGE_THREADS, default 1: with the observations in random order the
// pass is bound by the cache misses of lo / hi and by the scan's unpredictable range test -- 36 / 45 / 37 ms at 1 / 4 / 8
// threads on the same box; with the observations grouped by point it takes 5 ms serial)
int range_threads() {
  static const int v = [] {
    const char* e = getenv("GSLAM_HIP_BA_ORDER_RANGE_THREADS");
    const int t = e ? atoi(e) : 1;
    return t < 1 ? 1 : (t > 16 ? 16 : t);
  }();
  return v;
}

// ---------------------------------------------------------------- per-point camera range, on the host pool
// lo[p] / hi[p] = smallest / largest POSITION (pos[camera], or the camera index itself when pos is null) among the observers of
// point p (INT32_MAX / -1 for an unobserved point).  Every pool thread owns a contiguous RANGE OF POINTS and scans the whole
// observation list for them: the scan is sequential, the thread's part of lo / hi (8 MB / threads at C5) stays in its cache, no
// atomics, nothing to merge.  (Slices of the observation list with full-size arrays per thread were slower than one thread in
// the build container: eight 8 MB working sets fall out of the last-level cache.)  Serial by default: see range_threads.
// false: an index out of range (ba_run reports it).
bool point_ranges(const gh_ba_problem* pr, const int32_t* pos, std::vector<int32_t>& lo, std::vector<int32_t>& hi) {
  const int nc = pr->n_cams, np = pr->n_points, no = pr->n_obs;
  HostPool& pool = HostPool::get();
  const int T = no >= (1 << 18) ? std::min(pool.size(), range_threads()) : 1;
  lo.resize((size_t)np);
  hi.resize((size_t)np);
  std::vector<uint8_t> bad((size_t)T, 0);
  pool.run(T, [&](int t) {
    const int32_t p0 = (int32_t)((long long)np * t / T), p1 = (int32_t)((long long)np * (t + 1) / T);
    int32_t* const l = lo.data();
    int32_t* const h = hi.data();
    for (int32_t p = p0; p < p1; ++p) {
      l[p] = INT32_MAX;
      h[p] = -1;
    }
    const int32_t* const oc = pr->obs_cam;
    const int32_t* const op = pr->obs_point;
    uint8_t is_bad = 0;
    // A RUN of observations of one point (the usual order of an observation list) is folded in registers and written once: with the
    // list grouped by point the branch below is taken every k-th time (predictable) and lo / hi see one update per point; with the
    // list in random order every observation is its own run and the updates are the unconditional minimum / maximum -- no
    // compare-and-branch on the camera position, which mispredicts every other time when the cameras are in random order (23 ms
    // against 5 for C5's 6 M observations) and costs nothing but is slower when they are not (9 ms with unconditional stores).
    int32_t cur = -1, cl = INT32_MAX, ch = -1;
    for (int k = 0; k < no; ++k) {
      const int32_t p = op[k];
      if (p < p0 || p >= p1) {
        is_bad |= (uint8_t)((p < 0) | (p >= np));
        continue;
      }
      const int32_t c0 = oc[k];
      if (c0 < 0 || c0 >= nc) {
        is_bad = 1;
        continue;
      }
      const int32_t c = pos ? pos[c0] : c0;
      if (p != cur) {
        if (cur >= 0) {
          l[cur] = std::min(l[cur], cl);
          h[cur] = std::max(h[cur], ch);
        }
        cur = p;
        cl = ch = c;
      } else {
        cl = std::min(cl, c);
        ch = std::max(ch, c);
      }
    }
    if (cur >= 0) {
      l[cur] = std::min(l[cur], cl);
      h[cur] = std::max(h[cur], ch);
    }
    bad[t] = is_bad;
  });
  for (int t = 0; t < T; ++t)
    if (bad[t]) return false;
  return true;
}

// ---------------------------------------------------------------- arrow ordering (loop closures)
// The band solver needs every point's observers within kBandSpan camera positions of each other; ONE point seen from both
// ends of a loop used to send the whole graph to the dense factorisation (C5: 163 -> 0.89 LM iterations per second).  Here the
// few cameras such points tie to far-away ones are moved to the END of the camera order: for every long-range point the
// window of kBandSpan + 1 camera positions that holds most of its observers stays in the band, its other observers join the
// border.  What is left is a band (the band cameras are renumbered compactly: spans only shrink) + a dense border, the shape of
// chol_cr.hip's arrowhead solve -- what Ceres' SPARSE_SCHUR ordering achieves behind GSLAM/core/Optimizer.h:229, restated for
// trajectories.  pos (may be null = identity): position of every caller camera in the order the windows are measured in.
// Returns the number of border cameras (0: leave the order -- already a band, too many border cameras, or too few band cameras)
// and perm[new] = POSITION (in that order); *span_out = the largest distance, in positions, between two BAND observers of one
// point in the order that results (-1: not measured -- a bad index).
int arrow_order(const gh_ba_problem* pr, const int32_t* pos, std::vector<int32_t>& perm, int* span_out,
                std::vector<int32_t>* long_points = nullptr, int* span_short_out = nullptr) {
  // long_points (optional): the points whose observers lie more than kBandSpan positions apart (at most kMaxBorderPts + 1 are listed);
  // *span_short_out: the largest distance among the OTHER points -- the band's span when the long-range points stay out of it
  perm.clear();
  *span_out = -1;
  if (long_points) long_points->clear();
  const int nc = pr->n_cams, np = pr->n_points, no = pr->n_obs;
  if (np <= 0 || no <= 0) return 0;
  std::vector<int32_t> lo, hi;
  if (!point_ranges(pr, pos, lo, hi)) return 0;
  // the long-range points and their observers
  std::vector<int32_t> slot((size_t)np, -1);
  int nlong = 0, span = 0, span_short = 0;
  for (int p = 0; p < np; ++p) {
    if (hi[p] < 0) continue;
    const int d = hi[p] - lo[p];
    span = std::max(span, d);
    if (d > kBandSpan) {
      slot[p] = nlong++;
      if (long_points && (int)long_points->size() <= kMaxBorderPts) long_points->push_back(p);
    } else {
      span_short = std::max(span_short, d);
    }
  }
  *span_out = span;
  if (span_short_out) *span_short_out = span_short;
  if (nlong == 0 || nc < 4 * 32 + 1) return 0;
  // (a graph that is nowhere near a band has a long-range point for every few points; each of them puts at least one camera into
  //  the border: give up before building their lists)
  if (nlong > 4096 && (long long)nlong * 4 > (long long)np) return 0;
  std::vector<std::vector<int32_t>> seen((size_t)nlong);
  for (int k = 0; k < no; ++k) {
    const int32_t sl = slot[pr->obs_point[k]];
    if (sl >= 0) seen[sl].push_back(pos ? pos[pr->obs_cam[k]] : pr->obs_cam[k]);
  }
  std::vector<uint8_t> border((size_t)nc, 0);
  for (auto& v : seen) {
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    // cameras already in the border do not constrain the window
    size_t best_a = 0, best_cnt = 0;
    for (size_t a = 0, b = 0; a < v.size(); ++a) {
      if (border[v[a]]) continue;
      if (b < a) b = a;
      while (b + 1 < v.size() && v[b + 1] - v[a] <= kBandSpan) ++b;
      size_t cnt = 0;
      for (size_t t = a; t <= b; ++t) cnt += border[v[t]] ? 0 : 1;
      if (cnt > best_cnt) {
        best_cnt = cnt;
        best_a = a;
      }
    }
    for (size_t t = 0; t < v.size(); ++t)
      if (v[t] < v[best_a] || v[t] - v[best_a] > kBandSpan) border[v[t]] = 1;
  }
  int nb = 0;
  for (int c = 0; c < nc; ++c) nb += border[c];
  if (nb == 0 || nb > kMaxBorderCams || 6 * (nc - nb) < 4 * 64) return 0;
  perm.resize((size_t)nc);
  int w = 0;
  for (int c = 0; c < nc; ++c)
    if (!border[c]) perm[w++] = c;
  for (int c = 0; c < nc; ++c)
    if (border[c]) perm[w++] = c;
  // (the band part's span after the compact renumbering: at most kBandSpan by construction, at least what the short points have;
  //  ba_run measures it exactly on the renumbered graph)
  *span_out = -1;
  return nb;
}

// ---------------------------------------------------------------- bandwidth-reducing order of the co-visibility graph
// order[new] = old camera.  false: nothing to order (no observations).
bool covis_order(const gh_ba_problem* pr, std::vector<int32_t>& order) {
  const int nc = pr->n_cams, np = pr->n_points, no = pr->n_obs;
  if (nc < 2 || np < 1 || no < 2) return false;
  const bool timing = getenv("GSLAM_HIP_BA_TIMING") != nullptr;
  double t_prev = timing ? now_ms() : 0.0;
  auto lap = [&](const char* what) {
    if (!timing) return;
    const double t = now_ms();
    fprintf(stderr, "[gh_ba order] %s %.2f ms\n", what, t - t_prev);
    t_prev = t;
  };
  HostPool& pool = HostPool::get();
  // ---- the sample: the points whose index is a multiple of 2^sh -- about kSamplePerCam observations per camera, all of them when
  // the graph has fewer (a power of two: the test is a mask, not a division, in a pass over every observation)
  constexpr int kSamplePerCam = 48;
  int sh = 0;
  while (sh < 20 && ((long long)no >> (sh + 1)) >= (long long)kSamplePerCam * nc && ((long long)no >> (sh + 1)) >= (1 << 14)) ++sh;
  const int32_t mask = (1 << sh) - 1;
  const int nsp = ((np - 1) >> sh) + 1;  // sampled point s = p >> sh for p & mask == 0
  const int T = no >= (1 << 20) ? std::min(pool.size(), order_threads()) : 1;  // (waking the pool costs ~0.3 ms: C4's 300 k observations stay serial)
  std::vector<std::vector<int32_t>> tp((size_t)T), tc((size_t)T);
  pool.run(T, [&](int t) {
    const int k0 = (int)((long long)no * t / T), k1 = (int)((long long)no * (t + 1) / T);
    tp[t].reserve((size_t)((k1 - k0) >> sh) + 64);
    tc[t].reserve((size_t)((k1 - k0) >> sh) + 64);
    for (int k = k0; k < k1; ++k) {
      const int32_t p = pr->obs_point[k];
      if (p & mask) continue;
      tp[t].push_back(p >> sh);
      tc[t].push_back(pr->obs_cam[k]);
    }
  });
  std::vector<int32_t> pstart((size_t)nsp + 1, 0), cstart((size_t)nc + 1, 0);
  int ns = 0;
  for (int t = 0; t < T; ++t) {
    ns += (int)tp[t].size();
    for (size_t e = 0; e < tp[t].size(); ++e) {
      ++pstart[(size_t)tp[t][e] + 1];
      ++cstart[(size_t)tc[t][e] + 1];
    }
  }
  if (ns < 2) return false;
  for (int s = 0; s < nsp; ++s) pstart[(size_t)s + 1] += pstart[s];
  for (int c = 0; c < nc; ++c) cstart[(size_t)c + 1] += cstart[c];
  std::vector<int32_t> plist((size_t)ns), clist((size_t)ns), pfill(pstart.begin(), pstart.end() - 1), cfill(cstart.begin(), cstart.end() - 1);
  for (int t = 0; t < T; ++t)
    for (size_t e = 0; e < tp[t].size(); ++e) {
      const int32_t s = tp[t][e], c = tc[t][e];
      plist[(size_t)pfill[s]++] = c;  // cameras of sampled point s
      clist[(size_t)cfill[c]++] = s;  // sampled points of camera c
    }
  lap("sample lists");
  // ---- weighted adjacency, cameras in parallel (a point with very many observers would cost its square: only its first 64 take part)
  constexpr int kMaxObsPerPoint = 64;
  std::vector<int32_t> astart((size_t)nc + 1, 0), adj, wgt;
  {
    const int TA = ns >= (1 << 18) ? std::min(pool.size(), 2 * order_threads()) : 1;
    std::vector<std::vector<int32_t>> tadj((size_t)TA), twgt((size_t)TA);
    std::vector<int32_t> deg((size_t)nc, 0);
    pool.run(TA, [&](int t) {
      // camera ranges of equal sampled-observation count
      auto cam_at = [&](int part) {
        const int32_t target = (int32_t)((long long)ns * part / TA);
        return (int)(std::lower_bound(cstart.begin(), cstart.end(), target) - cstart.begin());
      };
      const int a0 = t == 0 ? 0 : std::min(cam_at(t), nc), a1 = t + 1 == TA ? nc : std::min(cam_at(t + 1), nc);
      std::vector<int32_t> mark((size_t)nc, -1), where((size_t)nc, 0);
      std::vector<int32_t>& A = tadj[t];
      std::vector<int32_t>& W = twgt[t];
      for (int a = a0; a < a1; ++a) {
        const size_t first = A.size();
        for (int32_t e = cstart[a]; e < cstart[(size_t)a + 1]; ++e) {
          const int32_t s = clist[e];
          const int32_t q1 = std::min(pstart[(size_t)s + 1], pstart[s] + kMaxObsPerPoint);
          for (int32_t q = pstart[s]; q < q1; ++q) {
            const int32_t b = plist[q];
            if (b == a) continue;
            if (mark[b] != a) {
              mark[b] = a;
              where[b] = (int32_t)A.size();
              A.push_back(b);
              W.push_back(1);
            } else {
              ++W[(size_t)where[b]];
            }
          }
        }
        deg[a] = (int32_t)(A.size() - first);
      }
    });
    for (int a = 0; a < nc; ++a) astart[(size_t)a + 1] = astart[a] + deg[a];
    adj.resize((size_t)astart[nc]);
    wgt.resize((size_t)astart[nc]);
    size_t w = 0;
    for (int t = 0; t < TA; ++t) {  // (the threads hold consecutive camera ranges: their lists concatenate)
      if (!tadj[t].empty()) {
        memcpy(&adj[w], tadj[t].data(), tadj[t].size() * sizeof(int32_t));
        memcpy(&wgt[w], twgt[t].data(), twgt[t].size() * sizeof(int32_t));
      }
      w += tadj[t].size();
    }
  }
  lap("adjacency");
  // ---- weighted maximum-adjacency order of every connected component.  The unnumbered camera with the largest total weight
  // to the numbered ones comes next: a bucket per key value (keys only grow; a stale entry is skipped when it surfaces).
  // The START matters: from the middle of a trajectory the numbered set grows both ways and neighbours end up two window
  // widths apart.  The LAST camera of any such order is an end of the trajectory (the side that is exhausted last), and unlike
  // the last level of a breadth-first sweep it is found by following the strong edges -- a few loop-closure points shortcut
  // breadth-first levels, they do not outweigh a camera's shared points with its neighbours.  So: order from the seed, then
  // again from the last camera of that order.
  std::vector<uint8_t> numbered((size_t)nc, 0);
  std::vector<int32_t> key((size_t)nc, 0), comp;
  std::vector<std::vector<int32_t>> bucket;
  auto max_adjacency = [&](int root, std::vector<int32_t>& out) {
    out.clear();
    int top = 0;
    if (bucket.empty()) bucket.resize(1);
    bucket[0].push_back(root);
    while (top >= 0) {
      if (bucket[top].empty()) {
        --top;
        continue;
      }
      const int v = bucket[top].back();
      bucket[top].pop_back();
      if (numbered[v] || key[v] != top) continue;  // (stale)
      numbered[v] = 1;
      out.push_back(v);
      for (int32_t e = astart[v]; e < astart[(size_t)v + 1]; ++e) {
        const int32_t b = adj[e];
        if (numbered[b]) continue;
        key[b] += wgt[e];
        if ((size_t)key[b] >= bucket.size()) bucket.resize((size_t)key[b] * 2 + 1);
        bucket[key[b]].push_back(b);
        if (key[b] > top) top = key[b];
      }
    }
  };
  order.clear();
  order.reserve((size_t)nc);
  std::vector<int32_t> pass;
  for (int seed = 0; seed < nc; ++seed) {
    if (numbered[seed] || astart[(size_t)seed + 1] == astart[seed]) continue;  // (unobserved in the sample: appended at the end)
    int root = seed;
    for (int sweep = 0; sweep < 2; ++sweep) {
      max_adjacency(root, pass);
      if (sweep == 1) break;
      root = pass.back();
      for (int32_t v : pass) {  // (un-number the component for the next sweep)
        numbered[v] = 0;
        key[v] = 0;
      }
    }
    order.insert(order.end(), pass.begin(), pass.end());
  }
  for (int c = 0; c < nc; ++c)
    if (!numbered[c]) order.push_back(c);  // cameras without a sampled observation (or without any): behind everything else
  lap("maximum-adjacency order");
  // ---- barycentre refinement on the sample (points and cameras in parallel on the pool: each writes only its own entry)
  std::vector<double> cpos((size_t)nc), ppos((size_t)nsp);
  std::vector<int32_t> rank((size_t)nc);
  constexpr int kMaxSweeps = 8;  // (C4 shuffled, span 24 in trajectory order: 0 / 1 / 2 / 4 / 8 sweeps leave spans of 45 / 36 / 32 / 28 / 27)
  const int TS = ns >= (1 << 18) ? std::min(pool.size(), 2 * order_threads()) : 1;
  std::vector<int32_t> before;
  for (int sweep = 0; sweep < kMaxSweeps; ++sweep) {
    before = order;
    for (int i = 0; i < nc; ++i) rank[order[i]] = i;
    pool.run(TS, [&](int t) {
      const int s0 = (int)((long long)nsp * t / TS), s1 = (int)((long long)nsp * (t + 1) / TS);
      for (int s = s0; s < s1; ++s) {
        const int32_t q0 = pstart[s], q1 = pstart[(size_t)s + 1];
        if (q0 == q1) continue;
        int32_t lo = INT32_MAX, hi = -1;
        double sum = 0.0;
        for (int32_t q = q0; q < q1; ++q) {
          const int32_t r = rank[plist[q]];
          lo = std::min(lo, r);
          hi = std::max(hi, r);
          sum += r;
        }
        ppos[s] = hi - lo > 4 * kBandSpan ? -1.0 : sum / (q1 - q0);  // (a long-range point does not vote)
      }
    });
    pool.run(TS, [&](int t) {
      const int c0 = (int)((long long)nc * t / TS), c1 = (int)((long long)nc * (t + 1) / TS);
      for (int c = c0; c < c1; ++c) {
        double sum = 0.0;
        int cnt = 0;
        for (int32_t e = cstart[c]; e < cstart[(size_t)c + 1]; ++e)
          if (ppos[clist[e]] >= 0.0) {
            sum += ppos[clist[e]];
            ++cnt;
          }
        cpos[c] = cnt ? sum / cnt : (double)rank[c];
      }
    });
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return cpos[a] < cpos[b]; });
    if (order == before) break;
  }
  // ---- orientation: a trajectory can be walked from either end.  The caller's observation list usually names a point's cameras in
  // the caller's camera order; keep that direction (the set-up kernels that sort per-camera lists like their input nearly sorted:
  // with the order reversed the first cost of a shuffled C5 took 20 ms instead of 0.1).  Vote of the sampled points: position of
  // the last listed camera against the first.
  {
    for (int i = 0; i < nc; ++i) rank[order[i]] = i;
    long long up = 0, down = 0;
    for (int s = 0; s < nsp; ++s) {
      const int32_t q0 = pstart[s], q1 = pstart[(size_t)s + 1];
      if (q1 - q0 < 2) continue;
      const int32_t d = rank[plist[q1 - 1]] - rank[plist[q0]];
      up += d > 0;
      down += d < 0;
    }
    const char* fe = getenv("GSLAM_HIP_BA_ORDER_FLIP");  // (experiments: the other direction)
    if ((down > up) != (fe && fe[0] == '1')) std::reverse(order.begin(), order.end());
  }
  lap("barycentre sweeps");
  return true;
}

}  // namespace

// perm[new] = old camera (empty: the caller's order stands), returns the number of border cameras at the end of the order.
// *reordered (may be null): 1 when the bandwidth-reducing order was applied.  `allow_reorder` = false: rounds 4-5 (A/B runs).
// The border of an arrowhead system can be made of the far CAMERAS of the long-range points (6 unknowns each; the cameras move to
// the end of the order) or of the long-range POINTS themselves (3 unknowns each: they are kept OUT of the Schur complement and
// solved for together with the cameras; no camera moves).  A revisit seen by a handful of cameras through hundreds of points wants
// the first, a handful of points seen from many far cameras the second: the smaller border wins.
static bool points_make_the_smaller_border(int nb_cams, const std::vector<int32_t>& long_pts, int nc) {
  const int np_b = (int)long_pts.size();
  if (np_b == 0 || np_b > kMaxBorderPts || nc < 4 * 32 + 1) return false;
  const char* e = getenv("GSLAM_HIP_BA_POINT_BORDER");  // "0": never (rounds 5's camera border, A/B); "1": whenever it is possible
  if (e && e[0] == '0') return false;
  if (e && e[0] == '1') return true;
  return nb_cams == 0 || 3 * np_b < 6 * nb_cams;
}

int gh_ba_order_cameras(const gh_ba_problem* pr, std::vector<int32_t>& perm, int* reordered, bool allow_reorder, int* band_span,
                        std::vector<int32_t>* border_points) {
  // *band_span (may be null): the camera span of the order that results when this function has measured it (no camera border), else -1
  // *border_points (may be null = never): filled when the long-range POINTS form the border (the return value is then 0)
  perm.clear();
  if (reordered) *reordered = 0;
  if (band_span) *band_span = -1;
  if (border_points) border_points->clear();
  const int nc = pr->n_cams;
  if (pr->n_points <= 0 || pr->n_obs <= 0) return 0;
  const bool timing = getenv("GSLAM_HIP_BA_TIMING") != nullptr;
  // a candidate: an order (empty = the caller's), its camera border / its long-range points, and what the border costs in unknowns
  struct Candidate {
    std::vector<int32_t> perm_arrow, long_pts;  // perm_arrow[new] = position (arrow ordering on top of the order), when nb > 0
    int nb = 0, span = -1, span_short = -1;
    bool points = false;
    long long cost = 0;  // border unknowns; -1: neither a band nor a band + border
  };
  auto evaluate = [&](const int32_t* pos, Candidate& c, const char* what) {
    const double t0 = timing ? now_ms() : 0.0;
    c.nb = arrow_order(pr, pos, c.perm_arrow, &c.span, border_points ? &c.long_pts : nullptr, &c.span_short);
    c.points = border_points != nullptr && points_make_the_smaller_border(c.nb, c.long_pts, nc);
    if (c.points) c.cost = 3LL * (long long)c.long_pts.size();
    else if (c.nb > 0) c.cost = 6LL * c.nb;
    else c.cost = (c.span >= 0 && c.span <= kBandSpan) ? 0 : -1;
    if (c.cost > 3LL * nc) {  // a border of more than half the system: the dense factorisation does the same work without the detour
      c.cost = -1;
      c.nb = 0;
      c.points = false;
    }
    if (timing)
      fprintf(stderr, "[gh_ba order] %s: span %d, %d border cameras, %zu long-range points -> %s, %.2f ms\n", what, c.span, c.nb, c.long_pts.size(),
              c.cost < 0 ? "no band" : (c.cost == 0 ? "a band" : (c.points ? "band + point border" : "band + camera border")), now_ms() - t0);
  };
  Candidate A;
  evaluate(nullptr, A, "caller's order");
  // The caller's order stands when it is a band, or a band with a border that is small next to the system (a trajectory with
  // loop closures handed over in order: no reason to pay for the ordering).  A LARGE border in the caller's order -- cameras in
  // random order can come out as "47 band cameras + 489 border cameras" -- is a sign that the order is the problem.
  const long long small = std::max<long long>(192, 6LL * nc / 16);
  Candidate B;
  std::vector<int32_t> order;
  bool have_b = false;
  if (allow_reorder && nc >= 4 * 32 + 1 && A.span > kBandSpan && (A.cost < 0 || A.cost > small) && covis_order(pr, order)) {
    std::vector<int32_t> pos((size_t)nc);
    for (int i = 0; i < nc; ++i) pos[order[i]] = i;
    evaluate(pos.data(), B, "new order");
    have_b = true;
  }
  // the better candidate: a band or band + border beats none; then the smaller border; the caller's order on a tie
  const bool use_b = have_b && B.cost >= 0 && (A.cost < 0 || B.cost < A.cost);
  if (!use_b && have_b && A.cost < 0 && B.span >= 0 && A.span >= 0 && B.span < A.span) {
    // (neither is a band: the narrower order still gives the dense solver a sparser system; never worse than what came in)
    perm.swap(order);
    if (reordered) *reordered = 1;
    if (band_span) *band_span = B.span;
    return 0;
  }
  Candidate& C = use_b ? B : A;
  if (use_b && reordered) *reordered = 1;
  if (C.cost < 0) {  // (the caller's order, no band: the dense solver; ba_run still wants the span it measured)
    if (band_span) *band_span = C.span;
    return 0;
  }
  if (C.points) {
    if (use_b) perm.swap(order);
    border_points->swap(C.long_pts);
    if (band_span) *band_span = C.span_short;
    return 0;
  }
  if (C.nb > 0) {
    if (use_b) {
      perm.resize((size_t)nc);
      for (int i = 0; i < nc; ++i) perm[i] = order[C.perm_arrow[i]];
    } else {
      perm.swap(C.perm_arrow);
    }
    if (band_span) *band_span = -1;
    return C.nb;
  }
  if (use_b) perm.swap(order);  // a band
  if (band_span) *band_span = C.span;
  return 0;
}

/* Host-only face of the above for tests and tools (no GPU needed): perm_out[n_cams] = old camera of every new position (the
 * identity when the caller's order stands); *n_border = cameras of the dense border at the end of the order; *cam_span = largest
 * distance in NEW positions between two band observers of one point (border cameras left out); *reordered = 1 when the
 * bandwidth-reducing order was applied; *n_border_points (NULL = report the camera-border choice only) = long-range points kept
 * out of the Schur complement as the border instead of cameras (then *n_border = 0 and *cam_span leaves those points out).
 * Returns GH_OK, or GH_ERR_ARG for null pointers / indices out of range. */
extern "C" gh_status gh_ba_camera_order(const gh_ba_problem* pr, int32_t* perm_out, int32_t* n_border, int32_t* cam_span,
                                        int32_t* reordered, int32_t* n_border_points) {
  if (!pr || !perm_out || pr->n_cams < 1 || (pr->n_obs > 0 && (!pr->obs_cam || !pr->obs_point))) return GH_ERR_ARG;
  const int nc = pr->n_cams;
  for (int k = 0; k < pr->n_obs; ++k)
    if (pr->obs_cam[k] < 0 || pr->obs_cam[k] >= nc || pr->obs_point[k] < 0 || pr->obs_point[k] >= pr->n_points) return GH_ERR_ARG;
  std::vector<int32_t> perm, bpts;
  int re = 0;
  const int nb = gh_ba_order_cameras(pr, perm, &re, true, nullptr, n_border_points ? &bpts : nullptr);
  if (perm.empty())
    for (int c = 0; c < nc; ++c) perm_out[c] = c;
  else
    memcpy(perm_out, perm.data(), (size_t)nc * sizeof(int32_t));
  if (n_border) *n_border = nb;
  if (reordered) *reordered = re;
  if (n_border_points) *n_border_points = (int32_t)bpts.size();
  if (cam_span) {
    std::vector<int32_t> pos((size_t)nc);
    for (int i = 0; i < nc; ++i) pos[perm_out[i]] = i;
    const int nband = nc - nb;
    std::vector<uint8_t> skip((size_t)std::max(pr->n_points, 1), 0);
    for (int32_t p : bpts) skip[p] = 1;
    std::vector<int32_t> lo((size_t)std::max(pr->n_points, 1), INT32_MAX), hi((size_t)std::max(pr->n_points, 1), -1);
    for (int k = 0; k < pr->n_obs; ++k) {
      const int32_t c = pos[pr->obs_cam[k]], p = pr->obs_point[k];
      if (c >= nband || skip[p]) continue;
      lo[p] = std::min(lo[p], c);
      hi[p] = std::max(hi[p], c);
    }
    int span = 0;
    for (int p = 0; p < pr->n_points; ++p)
      if (hi[p] >= 0) span = std::max(span, hi[p] - lo[p]);
    *cam_span = span;
  }
  return GH_OK;
}
