// Levenberg-Marquardt bundle adjustment inner loop on gfx950 (f64), behind gh_ba_solve / gh_ba_pnp.
//
// Drop-in for the solver behind GSLAM::Optimizer::optimize(BundleGraph&) (GSLAM/core/Optimizer.h:229,
// problem container :102-172, config :174-182) and optimizePnP (:202-207).  Mirrors oracle/ba_oracle.c
// operation for operation (same residual, Jacobians, Huber IRLS weight, damping, trust-region policy);
// parity is tolerance based (f64 GPU vs f64 CPU, libm vs ocml sin/cos, summation order).
//
// Per LM iteration (all HBM/gather-bound except the dense solve):
//   lin_points   1 thread / point over its CSR observation list   -> Hpp (3x3), g_p
//   lin_cams     1 wave / camera over its CSR list, fixed shuffle tree -> Hcc (6x6), g_c   (deterministic)
//   damp_points  (Hpp + D)^-1 closed form
//   schur        S = Hcc + D - sum_p W Hpp^-1 W^T :  deterministic mode: pair list sorted by destination 6x6
//                block, 1 wave / block, fixed butterfly (no atomics); fast mode: f64 atomics
//   potrf/potrs  chol.hip (v_mfma_f64_16x16x4_f64 trailing updates)
//   backsub      1 thread / point:  dp = Hpp^-1 (-g_p - sum W^T dc)
//   update+eval  retract poses (guarded SE3::exp, GSLAM/core/SE3.h:257-287), candidate robust cost and the
//                model decrease, fixed-order block reduction -> 2 doubles read back by the host loop.
// Jacobians are recomputed where needed instead of being stored (80 B gathered beats 144 B of W traffic).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <thread>

#include "common.h"
#include "cr_map.h"
#include "host_pool.h"

gh_status gh_potrf_dev_impl(gh_ctx* ctx, double* A, int n, int lda, int* info_dev, int extra_rows, double* dinv,
                            double* xwork, unsigned* flow_state, bool store_diag, bool state_ready);
size_t gh_potrf_flow_words(const gh_ctx* ctx, int n, int extra_rows);
size_t gh_potrf_flow_flag_words(int n, int extra_rows);
std::mutex& gh_potrf_flow_mutex(int device);
// chol_cr.hip: band solver (block cyclic reduction)
int gh_cr_tiles(int n, int hbw);
size_t gh_cr_dinv_doubles(int n, int T);
size_t gh_cr_panel_doubles(int n, int T);
gh_status gh_cr_solve_dev_impl(gh_ctx* ctx, double* A, int n, int lda, int T, double* dinv, double* W, double* x_dev,
                               int* info_dev, bool info_ready);
size_t gh_arrow_ws_doubles(const gh_ctx* ctx, int n_band, int T, int nbr);
int gh_cr_compact_lda(int n_band, int T, int nbr, int* brow);
gh_status gh_arrow_solve_dev_impl(gh_ctx* ctx, double* A, int n_band, int nbr, int lda, int T, double* dinv, double* W, double* bws,
                                  double* x_dev, int* info_dev, bool info_ready, bool allow_flow, const uint8_t* border_nz, bool compact);
int gh_ba_order_cameras(const gh_ba_problem* pr, std::vector<int32_t>& perm, int* reordered, bool allow_reorder, int* band_span,
                        std::vector<int32_t>* border_points);  // ba_order.hip
size_t gh_cr_border_symbolic_bytes(int n_band, int T, int nbr);
void gh_cr_border_symbolic(int n_band, int T, int nbr, const uint8_t* init, uint8_t* out);
gh_status gh_csr_build_dev(gh_ctx* ctx, const int32_t* keys_host, int n_items, int n_keys, int32_t* start_host, int32_t* list_host);
gh_status gh_potrs_bwd_dev_impl(gh_ctx* ctx, const double* L, int n, int lda, double* b, double* work,
                                const double* dinv, const double* yv, long long ystride, double* xh, int* info_dev,
                                bool xh_ready);

namespace {

constexpr double kMinDepth = 1e-9;

struct Obs {
  double r[2], w, s;
  double Jc[12], Jp[6];
};

__device__ __forceinline__ void quat_rotate(const double* q, const double* p, double* o) {
  double uvx = q[1] * p[2] - q[2] * p[1], uvy = q[2] * p[0] - q[0] * p[2], uvz = q[0] * p[1] - q[1] * p[0];
  uvx += uvx; uvy += uvy; uvz += uvz;
  o[0] = p[0] + q[3] * uvx + (q[1] * uvz - q[2] * uvy);
  o[1] = p[1] + q[3] * uvy + (q[2] * uvx - q[0] * uvz);
  o[2] = p[2] + q[3] * uvz + (q[0] * uvy - q[1] * uvx);
}

__device__ __forceinline__ void quat_mul(const double* a, const double* b, double* o) {
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}

// T <- T * exp([v, w]) with guarded coefficients (the reference's SE3::exp is NaN at w == 0)
__device__ void se3_retract(const double* pose, const double* xi, double* out) {
  const double* v = xi;
  const double* w = xi + 3;
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double th = sqrt(th2);
  double imag, real, A, B;
  if (th < 1e-5) {
    const double th4 = th2 * th2;
    imag = 0.5 - th2 / 48.0 + th4 / 3840.0;
    real = 1.0 - th2 / 8.0 + th4 / 384.0;
    A = 0.5 - th2 / 24.0 + th4 / 720.0;
    B = 1.0 / 6.0 - th2 / 120.0 + th4 / 5040.0;
  } else {
    imag = sin(0.5 * th) / th;
    real = cos(0.5 * th);
    A = (1.0 - cos(th)) / th2;
    B = (th - sin(th)) / (th2 * th);
  }
  double e[7];
  e[0] = imag * w[0]; e[1] = imag * w[1]; e[2] = imag * w[2]; e[3] = real;
  const double c1[3] = {w[1] * v[2] - w[2] * v[1], w[2] * v[0] - w[0] * v[2], w[0] * v[1] - w[1] * v[0]};
  const double c2[3] = {w[1] * c1[2] - w[2] * c1[1], w[2] * c1[0] - w[0] * c1[2], w[0] * c1[1] - w[1] * c1[0]};
  for (int i = 0; i < 3; ++i) e[4 + i] = v[i] + A * c1[i] + B * c2[i];
  double q[4], t[3];
  quat_mul(pose, e, q);
  quat_rotate(pose, e + 4, t);
  const double nrm = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) out[i] = q[i] * nrm;
  for (int i = 0; i < 3; ++i) out[4 + i] = pose[4 + i] + t[i];
}

// residual / weight / Jacobians of one observation; false if the point is not in front of the camera
template <bool WITH_J>
__device__ __forceinline__ bool linearize(const double* pose, int dof, const double* X, int pfree, const double* m,
                                          const double* info, double huber, Obs& o) {
  const double qc[4] = {-pose[0], -pose[1], -pose[2], pose[3]};
  const double d[3] = {X[0] - pose[4], X[1] - pose[5], X[2] - pose[6]};
  double Xc[3];
  quat_rotate(qc, d, Xc);
  if (!(Xc[2] > kMinDepth)) return false;
  const double iz = 1.0 / Xc[2];
  const double u = Xc[0] * iz, v = Xc[1] * iz;
  o.r[0] = u - m[0];
  o.r[1] = v - m[1];
  double L00 = 1, L01 = 0, L10 = 0, L11 = 1;
  if (info) { L00 = info[0]; L01 = info[1]; L10 = info[2]; L11 = info[3]; }
  const double s = o.r[0] * (L00 * o.r[0] + L01 * o.r[1]) + o.r[1] * (L10 * o.r[0] + L11 * o.r[1]);
  double w = 1.0;
  if (huber > 0 && s > huber * huber) w = huber / sqrt(s);
  o.w = w;
  o.s = s;
  if (!WITH_J) return true;
  const double P[6] = {iz, 0, -u * iz, 0, iz, -v * iz};
  const double D[18] = {-1, 0, 0, 0, -Xc[2], Xc[1],
                        0, -1, 0, Xc[2], 0, -Xc[0],
                        0, 0, -1, -Xc[1], Xc[0], 0};
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      double acc = 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) acc += P[a * 3 + j] * D[j * 6 + k];
      o.Jc[a * 6 + k] = ((dof >> k) & 1) ? acc : 0.0;
    }
  const double x = pose[0], y = pose[1], z = pose[2], qw = pose[3];
  const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - qw * z), 2 * (x * z + qw * y),
                       2 * (x * y + qw * z), 1 - 2 * (x * x + z * z), 2 * (y * z - qw * x),
                       2 * (x * z - qw * y), 2 * (y * z + qw * x), 1 - 2 * (x * x + y * y)};
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double acc = 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) acc += P[a * 3 + j] * R[k * 3 + j];  // R^T[j][k] = R[k][j]
      o.Jp[a * 3 + k] = pfree ? acc : 0.0;
    }
  return true;
}

__device__ __forceinline__ void weighted_info(const double* info, double w, double* L) {
  L[0] = w; L[1] = 0; L[2] = 0; L[3] = w;
  if (info) { L[0] = w * info[0]; L[1] = w * info[1]; L[2] = w * info[2]; L[3] = w * info[3]; }
}

struct Problem {
  int nc, np, no;
  const double* poses;   // nc x 7
  const int32_t* dof;
  const double* pts;     // np x 3
  const uint8_t* pfree;  // may be null
  const int32_t* ocam;
  const int32_t* opt;
  const double* oxy;
  const double* oinfo;   // may be null
  const int32_t* pstart; // CSR by point
  const int32_t* plist;
  const int32_t* cstart; // CSR by camera
  const int32_t* clist;
  double huber;
};

__device__ __forceinline__ bool lin_obs(const Problem& P, int k, Obs& o, bool with_j) {
  const int ci = P.ocam[k], pi = P.opt[k];
  const double* info = P.oinfo ? P.oinfo + 4 * k : nullptr;
  const int pf = P.pfree ? P.pfree[pi] : 1;
  if (with_j) return linearize<true>(P.poses + 7 * ci, P.dof[ci], P.pts + 3 * pi, pf, P.oxy + 2 * k, info, P.huber, o);
  return linearize<false>(P.poses + 7 * ci, P.dof[ci], P.pts + 3 * pi, pf, P.oxy + 2 * k, info, P.huber, o);
}

// ---------------------------------------------------------------- linearisation
__device__ __forceinline__ void lin_points_block(const Problem& P, double* __restrict__ Hpp, double* __restrict__ gp,
                                                 unsigned long long* __restrict__ gmax_bits, int bid) {
  const int p = bid * 256 + threadIdx.x;
  if (p >= P.np) return;
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
  for (int q = P.pstart[p]; q < P.pstart[p + 1]; ++q) {
    const int k = P.plist[q];
    Obs o;
    if (!lin_obs(P, k, o, true)) continue;
    double L[4];
    weighted_info(P.oinfo ? P.oinfo + 4 * k : nullptr, o.w, L);
    const double Lr[2] = {L[0] * o.r[0] + L[1] * o.r[1], L[2] * o.r[0] + L[3] * o.r[1]};
    double LJp[6];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      LJp[j] = L[0] * o.Jp[j] + L[1] * o.Jp[3 + j];
      LJp[3 + j] = L[2] * o.Jp[j] + L[3] * o.Jp[3 + j];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      g[a] += o.Jp[a] * Lr[0] + o.Jp[3 + a] * Lr[1];
#pragma unroll
      for (int b = 0; b < 3; ++b) H[3 * a + b] += o.Jp[a] * LJp[b] + o.Jp[3 + a] * LJp[3 + b];
    }
  }
  double gm = 0;
#pragma unroll
  for (int a = 0; a < 9; ++a) Hpp[(size_t)9 * p + a] = H[a];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    gp[(size_t)3 * p + a] = g[a];
    gm = fmax(gm, fabs(g[a]));
  }
  if (gm > 0) atomicMax(gmax_bits, (unsigned long long)__double_as_longlong(gm));
}

// W_i = Jc^T L Jp (6x3), WH_i = W_i Hpp^-1
__device__ __forceinline__ void make_W(const Obs& o, const double* L, double* W) {
  double LJp[6];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    LJp[j] = L[0] * o.Jp[j] + L[1] * o.Jp[3 + j];
    LJp[3 + j] = L[2] * o.Jp[j] + L[3] * o.Jp[3 + j];
  }
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) W[3 * a + b] = o.Jc[a] * LJp[b] + o.Jc[6 + a] * LJp[3 + b];
}

__device__ __forceinline__ void mul_WH(const double* W, const double* Hi, double* WH) {
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) WH[3 * a + b] = W[3 * a] * Hi[b] + W[3 * a + 1] * Hi[3 + b] + W[3 * a + 2] * Hi[6 + b];
}

// Camera-side linearisation in chunks of at most kCamChunk observations (one workgroup per chunk, so a camera
// that sees 20k points is as parallel as one that sees 20): each lane accumulates its share of Hcc (upper
// triangle) / g_c, a fixed butterfly sums the lanes, the four wave totals are added in a fixed order and the
// chunk total goes to `partial`; lin_cams_reduce_kernel adds a camera's chunks in order -> bitwise reproducible.
// Also stores W_k = Jc^T L Jp of every observation (zero when the point is behind the camera): the Schur
// product and the point back-substitution gather W instead of re-linearising.
constexpr int kCamChunk = 1024;

struct CamChunks {
  const int32_t* cam;    // chunk -> camera
  const int32_t* q0;     // chunk -> first position in clist
  const int32_t* first;  // camera -> first chunk (nc + 1)
  int nchunks;
};

__device__ __forceinline__ void lin_cams_block(const Problem& P, const CamChunks& C, double* __restrict__ partial,
                                               double* __restrict__ Wbuf, int bid) {
  __shared__ double part[4][27];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = C.cam[bid];
  const int q_begin = C.q0[bid];
  const int q_end = min(q_begin + kCamChunk, P.cstart[c + 1]);
  double H[21], g[6];  // upper triangle, row-major packed
#pragma unroll
  for (int a = 0; a < 21; ++a) H[a] = 0;
#pragma unroll
  for (int a = 0; a < 6; ++a) g[a] = 0;
  for (int q = q_begin + threadIdx.x; q < q_end; q += 256) {
    const int k = P.clist[q];
    Obs o;
    double W[18];
    const bool ok = lin_obs(P, k, o, true);
    if (ok) {
      double L[4];
      weighted_info(P.oinfo ? P.oinfo + 4 * k : nullptr, o.w, L);
      make_W(o, L, W);
      const double Lr[2] = {L[0] * o.r[0] + L[1] * o.r[1], L[2] * o.r[0] + L[3] * o.r[1]};
      double LJc[12];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        LJc[j] = L[0] * o.Jc[j] + L[1] * o.Jc[6 + j];
        LJc[6 + j] = L[2] * o.Jc[j] + L[3] * o.Jc[6 + j];
      }
      int t = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        g[a] += o.Jc[a] * Lr[0] + o.Jc[6 + a] * Lr[1];
#pragma unroll
        for (int b = a; b < 6; ++b) H[t++] += o.Jc[a] * LJc[b] + o.Jc[6 + a] * LJc[6 + b];
      }
    } else {
#pragma unroll
      for (int t = 0; t < 18; ++t) W[t] = 0.0;
    }
    double2* dst = reinterpret_cast<double2*>(Wbuf + (size_t)18 * k);
#pragma unroll
    for (int t = 0; t < 9; ++t) dst[t] = make_double2(W[2 * t], W[2 * t + 1]);
  }
  // fixed-order butterfly: every lane ends with the same total, summed in the same order each run
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 21; ++a) H[a] += __shfl_xor(H[a], off);
#pragma unroll
    for (int a = 0; a < 6; ++a) g[a] += __shfl_xor(g[a], off);
  }
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 21; ++a) part[wv][a] = H[a];
#pragma unroll
    for (int a = 0; a < 6; ++a) part[wv][21 + a] = g[a];
  }
  __syncthreads();
  if (threadIdx.x < 27)
    partial[(size_t)27 * bid + threadIdx.x] =
        (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// ONE launch for both linearisations: the first `nchunks` workgroups take the camera chunks (the long ones: dispatched first),
// the rest a block of 256 points each.  The two do not depend on each other, and 196 point workgroups alone do not fill
// 256 CUs: as its own launch the point side cost 24 us per iteration plus a launch gap, here it runs beside the chunks.
__global__ __launch_bounds__(256) void lin_kernel(Problem P, CamChunks C, double* __restrict__ partial,
                                                  double* __restrict__ Wbuf, double* __restrict__ Hpp,
                                                  double* __restrict__ gp, unsigned long long* __restrict__ gmax_bits) {
  if ((int)blockIdx.x < C.nchunks) lin_cams_block(P, C, partial, Wbuf, (int)blockIdx.x);
  else lin_points_block(P, Hpp, gp, gmax_bits, (int)blockIdx.x - C.nchunks);
}

// 32 threads per camera: thread t < 27 adds element t of the camera's chunk totals in chunk order
__device__ __forceinline__ void lin_cams_reduce_block(int nc, const CamChunks& C, const double* __restrict__ partial,
                                                      double* __restrict__ Hcc, double* __restrict__ gc,
                                                      unsigned long long* __restrict__ gmax_bits, int bid) {
  const int c = bid * 8 + (threadIdx.x >> 5), t = threadIdx.x & 31;
  if (c >= nc || t >= 27) return;
  double tot = 0;
  for (int ch = C.first[c]; ch < C.first[c + 1]; ++ch) tot += partial[(size_t)27 * ch + t];
  if (t >= 21) {
    gc[(size_t)6 * c + (t - 21)] = tot;
    const double gm = fabs(tot);
    if (gm > 0) atomicMax(gmax_bits, (unsigned long long)__double_as_longlong(gm));
  } else {
    int a = 0, rem = t;  // packed upper triangle index -> (a, b), b >= a
    while (rem >= 6 - a) {
      rem -= 6 - a;
      ++a;
    }
    const int b = a + rem;
    Hcc[(size_t)36 * c + 6 * a + b] = tot;
    Hcc[(size_t)36 * c + 6 * b + a] = tot;
  }
}

__device__ __forceinline__ double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ void damp_points_block(int np, const double* __restrict__ Hpp, double radius,
                                                  double* __restrict__ Hpi, int* __restrict__ bad, int bid) {
  const int p = bid * 256 + threadIdx.x;
  if (p >= np) return;
  double H[9];
#pragma unroll
  for (int a = 0; a < 9; ++a) H[a] = Hpp[(size_t)9 * p + a];
#pragma unroll
  for (int a = 0; a < 3; ++a) H[4 * a] += clampd(H[4 * a], 1e-6, 1e32) / radius;
  const double a = H[0], b = H[1], c = H[2], d = H[4], e = H[5], f = H[8];
  const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const double det = a * c00 + b * c01 + c * c02;
  if (!(det > 0)) {
    atomicOr(bad, 1);
    return;
  }
  const double id = 1.0 / det;
  double* o = Hpi + (size_t)9 * p;
  o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
  o[3] = o[1]; o[4] = (a * f - c * c) * id; o[5] = (b * c - a * e) * id;
  o[6] = o[2]; o[7] = o[5]; o[8] = (a * d - b * b) * id;
}

__global__ __launch_bounds__(256) void damp_points_kernel(int np, const double* __restrict__ Hpp, double radius,
                                                          double* __restrict__ Hpi, int* __restrict__ bad) {
  damp_points_block(np, Hpp, radius, Hpi, bad, (int)blockIdx.x);
}
// after a fresh linearisation: the camera-side reduction (first `nrb` workgroups) and the point-side damping + inversion
// side by side -- neither reads what the other writes, and each alone leaves most of the CUs idle
__global__ __launch_bounds__(256) void lin_reduce_damp_kernel(int nc, CamChunks C, const double* __restrict__ partial,
                                                              double* __restrict__ Hcc, double* __restrict__ gc,
                                                              unsigned long long* __restrict__ gmax_bits, int nrb, int np,
                                                              const double* __restrict__ Hpp, double radius,
                                                              double* __restrict__ Hpi, int* __restrict__ bad) {
  if ((int)blockIdx.x < nrb) lin_cams_reduce_block(nc, C, partial, Hcc, gc, gmax_bits, (int)blockIdx.x);
  else damp_points_block(np, Hpp, radius, Hpi, bad, (int)blockIdx.x - nrb);
}

// S diagonal blocks (+ damping) and rhs = -g_c.  S is n x n column-major, zeroed beforehand.
__global__ __launch_bounds__(64) void schur_diag_kernel(int nc, const double* __restrict__ Hcc,
                                                        const double* __restrict__ gc, double radius,
                                                        double* __restrict__ S, int n, double* __restrict__ rhs) {
  const int c = blockIdx.x, t = threadIdx.x;
  if (t < 36) {
    const int a = t / 6, b = t - 6 * a;
    double v = Hcc[(size_t)36 * c + t];
    if (a == b) v += clampd(v, 1e-6, 1e32) / radius;
    S[(size_t)(6 * c + b) * n + 6 * c + a] = v;
  } else if (t < 42) {
    rhs[6 * c + (t - 36)] = -gc[6 * c + (t - 36)];
  }
}

// The same plus the clearing of S, in one pass over what a factorisation reads: for column c the rows from the top of
// its 64 x 64 diagonal tile down to the right-hand-side row n (pitch lda), i.e. the lower block triangle with whole
// diagonal tiles -- half the bytes of a memset of the full square, and two launches less.  Element (r, c) of a camera's
// diagonal 6 x 6 block gets H_cc (+ damping), row n gets rhs = -g_c (it rides through the factorisation), the rest zero.
__device__ __forceinline__ void schur_init_block(int n, int lda, const double* __restrict__ Hcc,
                                                 const double* __restrict__ gc, double radius, double* __restrict__ S,
                                                 double* __restrict__ rhs, int bx, int by) {
  // one workgroup = 2048 rows of one column (8 per thread: four 16-byte stores), starting at the column's diagonal tile
  const int c = by, r0 = (c >> 6) << 6;
  const int rb = r0 + bx * 2048;
  if (rb > n) return;
  const int cam = c / 6, b = c - 6 * cam;
  double* col = S + (size_t)c * lda;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int r = rb + 512 * e + 2 * threadIdx.x;  // r even, lda even: a 16-byte aligned pair inside the column
    if (r > n) break;
    double v[2] = {0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int rr = r + u;
      if (rr == n) {
        v[u] = -gc[c];
        rhs[c] = v[u];
      } else if (rr < n && rr / 6 == cam) {
        const int a = rr - 6 * cam;
        double h = Hcc[(size_t)36 * cam + 6 * a + b];
        if (a == b) h += clampd(h, 1e-6, 1e32) / radius;
        v[u] = h;
      }
    }
    if (r + 1 < lda) *reinterpret_cast<double2*>(col + r) = make_double2(v[0], v[1]);
    else col[r] = v[0];
  }
}

// The same for the BAND solver (chol_cr.hip), which only ever touches, in column c of superblock J = c / m: the band rows
// (from the top of the column's diagonal tile to the end of superblock J + 1) and the fill blocks B(J + 2 s, J) of the
// levels s at which J survives (J % 2 s == 0): ~ (2 + log2 N) m rows instead of n - c -- at C5 0.9 GB of zeros per LM
// iteration instead of 14.4 GB.  One workgroup per column.
// ARROWHEAD systems (nband < n: the last n - nband unknowns are the dense border of chol_cr.hip's arrowhead mode): a band column
// also clears its border rows nband .. n - 1, a border column its whole lower part.
// POINT BORDER of an arrowhead system (ba_order.hip: the long-range points make the smaller border): n_pts points stay out of the
// Schur complement; their 3 n_pts unknowns follow the 6 nc camera unknowns.  pts[b] = point of border slot b, slot[p] = b or -1.
struct PointBorder {
  int n_pts = 0;
  const int32_t* pts = nullptr;
  const int32_t* slot = nullptr;
  const double* Hpp = nullptr;  // 9 per point (undamped)
  const double* gp = nullptr;   // 3 per point
};

__device__ __forceinline__ void schur_init_band_block(int n, int lda, int m, const double* __restrict__ Hcc,
                                                      const double* __restrict__ gc, double radius, double* __restrict__ S,
                                                      double* __restrict__ rhs, int c, int nband, CrMap map, PointBorder pb) {
  // (map: cr_map.h -- dense columns of n + 1 rows, or compact columns that hold exactly the rows cleared here)
  const int cam = c / 6, b = c - 6 * cam;
  double* const colB = S + (size_t)c * lda + map.bshift();  // border rows and the right-hand-side row of this column
  if (c >= nband && pb.n_pts > 0) {
    // a column of a border POINT: its damped 3 x 3 block H_pp (what damp_points_block inverts for the other points) on the diagonal
    // of the corner, zeros elsewhere, right-hand side - g_p
    const int bslot = (c - nband) / 3, j = (c - nband) - 3 * bslot, p = pb.pts[bslot];
    const int r_first = map.m ? nband : ((c >> 6) << 6);
    for (int r = r_first + (int)threadIdx.x; r < n; r += 256) {
      double v = 0.0;
      if (r >= nband && (r - nband) / 3 == bslot) {
        const int i = (r - nband) - 3 * bslot;
        double h = pb.Hpp[(size_t)9 * p + 3 * i + j];
        if (i == j) h += clampd(h, 1e-6, 1e32) / radius;
        v = h;
      }
      (r >= nband ? colB : S + (size_t)c * lda)[r] = v;
    }
    if (threadIdx.x == 0) {
      const double v = -pb.gp[(size_t)3 * p + j];
      colB[n] = v;
      rhs[c] = v;
    }
    return;
  }
  if (c >= nband) {  // a border column: dense from its 64-row tile down (compact: its border rows)
    const int r_first = map.m ? nband : ((c >> 6) << 6);
    for (int r = r_first + (int)threadIdx.x; r < n; r += 256) {
      double v = 0.0;
      if (r / 6 == cam) {
        const int a = r - 6 * cam;
        double h = Hcc[(size_t)36 * cam + 6 * a + b];
        if (a == b) h += clampd(h, 1e-6, 1e32) / radius;
        v = h;
      }
      (r >= nband ? colB : S + (size_t)c * lda)[r] = v;
    }
    if (threadIdx.x == 0) {
      const double v = -gc[c];
      colB[n] = v;
      rhs[c] = v;
    }
    return;
  }
  for (int r = nband + (int)threadIdx.x; r < n; r += 256) colB[r] = 0.0;  // (nothing when there is no border)
  const int n_full = n;
  n = nband;
  const int J = c / m, N = (n + m - 1) / m;
  double* const col = S + (size_t)c * lda + map.shift(J, J);  // rows of the superblocks J and J + 1 (the same shift)
  const int r0 = (c >> 6) << 6, r1 = min(n, (J + 2) * m);
  for (int r = r0 + (int)threadIdx.x; r < r1; r += 256) {
    double v = 0.0;
    if (r / 6 == cam) {
      const int a = r - 6 * cam;
      double h = Hcc[(size_t)36 * cam + 6 * a + b];
      if (a == b) h += clampd(h, 1e-6, 1e32) / radius;
      v = h;
    }
    col[r] = v;
  }
  // (compact columns have a slot for the levels the reduction RUNS -- the dense top takes over at stride 2^levels; a fill block of
  //  a level beyond that would land in the border rows and the right-hand side)
  for (int s = 1; J + 2 * s < N && (map.m == 0 || 2 * s <= (1 << map.levels)); s *= 2) {
    if (J % (2 * s) != 0) break;  // (J survives level s only if it survived every level before)
    const int f0 = (J + 2 * s) * m, f1 = min(n, f0 + m);
    double* const colF = S + (size_t)c * lda + map.shift(J + 2 * s, J);
    for (int r = f0 + (int)threadIdx.x; r < f1; r += 256) colF[r] = 0.0;
  }
  if (threadIdx.x == 0) {
    const double v = -gc[c];
    colB[n_full] = v;
    rhs[c] = v;
  }
}

__global__ __launch_bounds__(256) void schur_init_kernel(int n, int lda, const double* __restrict__ Hcc,
                                                         const double* __restrict__ gc, double radius,
                                                         double* __restrict__ S, double* __restrict__ rhs, int band_m, int nband,
                                                         int gx, CrMap map, PointBorder pb) {
  // a one-dimensional grid of gx workgroups per column (gridDim.y stops at 65535: the limit n < 65536 of rounds 1-5 was this launch)
  const int b = (int)blockIdx.x;
  if (band_m) schur_init_band_block(n, lda, band_m, Hcc, gc, radius, S, rhs, b, nband, map, pb);  // (gx == 1)
  else schur_init_block(n, lda, Hcc, gc, radius, S, rhs, b % gx, b / gx);
}

__device__ __forceinline__ void load_W(const double* __restrict__ Wbuf, int k, double* W) {
  const double2* src = reinterpret_cast<const double2*>(Wbuf + (size_t)18 * k);
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const double2 v = src[t];
    W[2 * t] = v.x;
    W[2 * t + 1] = v.y;
  }
}

// element (r, c) of the reduced camera system, r >= c, as the assembly addresses it: a band row (at most one superblock below
// the column's: the camera span guarantees it), a border row, or the right-hand-side row (cr_map.h)
__device__ __forceinline__ double* s_elem(double* __restrict__ S, int lda, const CrMap& map, int r, int c) {
  if (map.m == 0) return S + (size_t)c * lda + r;
  return S + (size_t)c * lda + r + (r >= map.n_band ? map.bshift() : -(long long)(c / map.m) * map.m);
}
// Fast mode: one thread per observation i (in point-CSR order); all j of the same point; f64 atomics.
__global__ __launch_bounds__(256) void schur_atomic_kernel(Problem P, const double* __restrict__ Hpi,
                                                           const double* __restrict__ gp, double* __restrict__ S,
                                                           int n, double* __restrict__ rhs,
                                                           const double* __restrict__ Wbuf, CrMap map,
                                                           const int32_t* __restrict__ pbslot) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= P.no) return;
  const int k = P.plist[q];
  const int p = P.opt[k], ci = P.ocam[k];
  if (pbslot != nullptr && pbslot[p] >= 0) return;  // (a border point stays out of the Schur complement)
  double Wi[18], WH[18];
  load_W(Wbuf, k, Wi);
  mul_WH(Wi, Hpi + (size_t)9 * p, WH);
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    const double v = WH[3 * a] * gp[3 * p] + WH[3 * a + 1] * gp[3 * p + 1] + WH[3 * a + 2] * gp[3 * p + 2];
    atomicAdd(&rhs[6 * ci + a], v);
  }
  for (int q2 = P.pstart[p]; q2 < P.pstart[p + 1]; ++q2) {
    const int k2 = P.plist[q2];
    const int cj = P.ocam[k2];
    if (cj > ci) continue;  // only the lower triangle is stored; (j, i) covers the mirror block
    double Wj[18];
    load_W(Wbuf, k2, Wj);
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        if (ci == cj && a < b) continue;
        const double v = WH[3 * a] * Wj[3 * b] + WH[3 * a + 1] * Wj[3 * b + 1] + WH[3 * a + 2] * Wj[3 * b + 2];
        atomicAdd(s_elem(S, n, map, 6 * ci + a, 6 * cj + b), -v);
      }
  }
}

// Deterministic mode: segmented reduction over a pair list sorted by destination block.
// The host builds, once per solve (the structure is fixed across LM iterations), every contributing pair
// of observations (k, k2) of a common point with cam(k) >= cam(k2), stably sorted by (cam(k), cam(k2)),
// plus the list of distinct blocks.  One wave per block: lanes stride over the block's pairs, accumulate
// W_i Hpp^-1 W_j^T in registers, and a fixed butterfly sums the 64 partials -> bitwise reproducible,
// no atomics, and a camera with 20k observations is as parallel as one with 20.
constexpr int kSchurSeg = 256;  // pairs per wave

struct SchurBlocks {
  const int32_t* pair_a;   // observation index k  (row camera)
  const int32_t* pair_b;   // observation index k2 (column camera)
  const int32_t* bstart;   // nblocks + 1
  const int32_t* bci;
  const int32_t* bcj;
  const int32_t* seg_blk;    // segment -> block (a block is cut into segments of kSchurSeg pairs)
  const int32_t* seg_first;  // block -> first segment (nblocks + 1)
  int nblocks, nsegs;
};

// one step of the reduce-scatter butterfly of schur_blocks_block: K values per lane -> ceil(K / 2); lanes with bit M clear
// keep the lower half (and send the upper), the others the upper half (zero padded)
template <int K, int M>
__device__ __forceinline__ void schur_scatter_step(double (&v)[42], int lane, int& elem, int& mylen) {
  constexpr int H = (K + 1) / 2;
  const bool up = (lane & M) != 0;
#pragma unroll
  for (int j = 0; j < H; ++j) {
    const double lo = v[j], hi = H + j < K ? v[H + j] : 0.0;
    const double recv = __shfl_xor(up ? lo : hi, M);
    v[j] = (up ? hi : lo) + recv;
  }
  elem += up ? H : 0;
  mylen = up ? (mylen > H ? mylen - H : 0) : (mylen < H ? mylen : H);
}

// one wave per segment: partial[seg][0..35] = sum W_i Hpp^-1 W_j^T (row-major a, b), [36..41] = sum W_i Hpp^-1 g_p
__device__ __forceinline__ void schur_blocks_block(const Problem& P, const SchurBlocks& B, const double* __restrict__ Hpi,
                                                   const double* __restrict__ gp, const double* __restrict__ Wbuf,
                                                   double* __restrict__ partial, unsigned bid, unsigned nbid) {
  const int lane = threadIdx.x & 63;
  // `nbid` (a multiple of 8) workgroups take the segments; workgroup L runs on XCD L % 8: every XCD takes one contiguous eighth of the
  // segments, so that the blocks of one row camera (consecutive in the list) gather its W rows through one L2
  const int wg = (int)(bid & 7u) * (int)(nbid >> 3) + (int)(bid >> 3);
  const int seg = wg * 4 + (threadIdx.x >> 6);
  if (seg >= B.nsegs) return;
  const int blk = B.seg_blk[seg];
  const int e_begin = B.bstart[blk] + (seg - B.seg_first[blk]) * kSchurSeg;
  const int e_end = min(e_begin + kSchurSeg, B.bstart[blk + 1]);
  double acc[36], racc[6];
#pragma unroll
  for (int t = 0; t < 36; ++t) acc[t] = 0;
#pragma unroll
  for (int t = 0; t < 6; ++t) racc[t] = 0;
  // (the pair indices of the NEXT trip are asked for before this trip's blocks: one level less in the chain of dependent
  //  loads index -> point / W blocks -> point block that bounds this kernel at two waves per SIMD)
  int e = e_begin + lane;
  int k = e < e_end ? B.pair_a[e] : -1, k2 = e < e_end ? B.pair_b[e] : -1;
  for (; k >= 0; e += 64) {
    const int en = e + 64;
    const int kn = en < e_end ? B.pair_a[en] : -1, k2n = en < e_end ? B.pair_b[en] : -1;
    const int p = P.opt[k];
    double Wi[18], WH[18], Wj[18];
    load_W(Wbuf, k, Wi);
    load_W(Wbuf, k2, Wj);  // zero for an observation behind its camera: the pair contributes nothing
    mul_WH(Wi, Hpi + (size_t)9 * p, WH);
    if (k2 == k) {
#pragma unroll
      for (int a = 0; a < 6; ++a)
        racc[a] += WH[3 * a] * gp[3 * p] + WH[3 * a + 1] * gp[3 * p + 1] + WH[3 * a + 2] * gp[3 * p + 2];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b)
        acc[6 * a + b] += WH[3 * a] * Wj[3 * b] + WH[3 * a + 1] * Wj[3 * b + 1] + WH[3 * a + 2] * Wj[3 * b + 2];
    k = kn;
    k2 = k2n;
  }
  // Reduce-scatter butterfly over the wave: at every step a lane keeps one half of its values and trades the other half
  // with its partner, so the 42 sums cost 21 + 11 + 6 + 3 + 2 + 1 = 44 exchanges instead of 42 x 6, and every lane ends up
  // with (at most) ONE element's total: element 21 b5 + 11 b4 + 6 b3 + 3 b2 + 2 b1 + b0 of its lane number's bits.
  // Fixed order of additions: deterministic, like the butterfly it replaces.
  double v[42];
#pragma unroll
  for (int t = 0; t < 36; ++t) v[t] = acc[t];
#pragma unroll
  for (int t = 0; t < 6; ++t) v[36 + t] = racc[t];
  int elem = 0, mylen = 42;
  schur_scatter_step<42, 32>(v, lane, elem, mylen);
  schur_scatter_step<21, 16>(v, lane, elem, mylen);
  schur_scatter_step<11, 8>(v, lane, elem, mylen);
  schur_scatter_step<6, 4>(v, lane, elem, mylen);
  schur_scatter_step<3, 2>(v, lane, elem, mylen);
  schur_scatter_step<2, 1>(v, lane, elem, mylen);
  if (mylen > 0) partial[(size_t)42 * seg + elem] = v[0];
}

__global__ __launch_bounds__(256) void schur_blocks_kernel(Problem P, SchurBlocks B, const double* __restrict__ Hpi,
                                                           const double* __restrict__ gp,
                                                           const double* __restrict__ Wbuf,
                                                           double* __restrict__ partial) {
  schur_blocks_block(P, B, Hpi, gp, Wbuf, partial, blockIdx.x, gridDim.x);
}
// The segment sums (first `nsb` workgroups, a multiple of 8) and the seeding of S with the damped camera blocks + the
// right-hand side row, in ONE launch: the sums only write `partial`, the seed only S; schur_reduce_kernel, which joins them,
// comes behind both either way.  `gx` = workgroups per column of the seed.
__global__ __launch_bounds__(256) void schur_blocks_init_kernel(Problem P, SchurBlocks B, const double* __restrict__ Hpi,
                                                                const double* __restrict__ gp,
                                                                const double* __restrict__ Wbuf,
                                                                double* __restrict__ partial, unsigned nsb, int n, int lda,
                                                                const double* __restrict__ Hcc,
                                                                const double* __restrict__ gc, double radius,
                                                                double* __restrict__ S, double* __restrict__ rhs, int gx,
                                                                int band_m, int nband, CrMap map, PointBorder pb) {
  if (blockIdx.x < nsb) {
    schur_blocks_block(P, B, Hpi, gp, Wbuf, partial, blockIdx.x, nsb);
  } else {
    const int b = (int)(blockIdx.x - nsb);
    if (band_m) schur_init_band_block(n, lda, band_m, Hcc, gc, radius, S, rhs, b, nband, map, pb);  // (gx == 1)
    else schur_init_block(n, lda, Hcc, gc, radius, S, rhs, b % gx, b / gx);
  }
}

// block-wide exclusive scan of one int per thread (256 threads); returns the exclusive prefix, *total = the sum
__device__ __forceinline__ int block_excl_scan(int v, int* wave_tot /* shared[4] */, int* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  __syncthreads();  // protect wave_tot reuse
  if (lane == 63) wave_tot[wv] = incl;
  __syncthreads();
  int off = 0;
  for (int k = 0; k < wv; ++k) off += wave_tot[k];
  *total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
  return off + incl - v;
}

// ---------------------------------------------------------------------------------------------------------------
// The pair list of the deterministic Schur product, built on the GPU (large graphs: on the host it was the largest part of
// the per-solve setup, 1.6 ms on 16 threads + 8 MB of upload at C4).  Same lists as build_schur_pairs below, element for
// element (tests compare them):
//   generation order = cameras ascending, the camera's observations k ascending, the point's observations k2 ascending,
//   a pair is kept if cam(k2) <= cam(k);  then a STABLE sort by cam(k2) inside each camera  ->  blocks (ci, cj) ascending
//   with the generation order kept inside a block.
// count -> exclusive scan -> emit -> one workgroup per camera sorts its pairs in LDS (bitonic on (cj << 32 | position):
// the position makes it stable) and marks block starts -> scan over cameras -> block tables -> scan -> segment tables.
constexpr int kPairSortMaxCams = 15360;  // one LDS cursor per partner camera (60 KB); more cameras -> host path

__global__ __launch_bounds__(256) void pair_count_kernel(int no, const int32_t* __restrict__ clist,
                                                         const int32_t* __restrict__ ocam, const int32_t* __restrict__ opt,
                                                         const int32_t* __restrict__ pstart,
                                                         const int32_t* __restrict__ plist, int32_t* __restrict__ cnt,
                                                         const int32_t* __restrict__ pbslot) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= no) return;
  const int k = clist[q], ci = ocam[k], p = opt[k];
  if (pbslot != nullptr && pbslot[p] >= 0) {  // a border point stays out of the Schur complement: no pairs
    cnt[q] = 0;
    return;
  }
  int c = 0;
  for (int q2 = pstart[p]; q2 < pstart[p + 1]; ++q2) c += ocam[plist[q2]] <= ci ? 1 : 0;
  cnt[q] = c;
}

// exclusive scan of n ints, 1024 per workgroup: out[i] = sum of in[< i] within the workgroup's chunk, sums[blk] = chunk total
__global__ __launch_bounds__(256) void scan_chunks_kernel(const int32_t* __restrict__ in, int n, int32_t* __restrict__ out,
                                                          int32_t* __restrict__ sums) {
  __shared__ int wave_tot[4];
  const int base = blockIdx.x * 1024 + 4 * threadIdx.x;
  int v[4], t = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[e] = base + e < n ? in[base + e] : 0;
    t += v[e];
  }
  int total;
  int run = block_excl_scan(t, wave_tot, &total);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (base + e < n) out[base + e] = run;
    run += v[e];
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
// one workgroup: sums[0 .. m) -> exclusive prefix in place, grand total to sums[m] and *total_out
__global__ __launch_bounds__(256) void scan_sums_kernel(int32_t* __restrict__ sums, int m, int32_t* __restrict__ total_out) {
  __shared__ int wave_tot[4];
  int carry = 0;
  for (int b0 = 0; b0 < m; b0 += 256) {
    const int i = b0 + threadIdx.x;
    const int v = i < m ? sums[i] : 0;
    int total;
    const int ex = block_excl_scan(v, wave_tot, &total);
    if (i < m) sums[i] = carry + ex;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sums[m] = carry;
    if (total_out) *total_out = carry;
  }
}
__global__ __launch_bounds__(256) void scan_add_kernel(int32_t* __restrict__ out, int n, const int32_t* __restrict__ sums,
                                                       int m) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] += sums[i >> 10];
  if (i == 0) out[n] = sums[m];  // out has n + 1 entries: the total closes it
}

__global__ __launch_bounds__(256) void pair_emit_kernel(int no, const int32_t* __restrict__ clist,
                                                        const int32_t* __restrict__ ocam, const int32_t* __restrict__ opt,
                                                        const int32_t* __restrict__ pstart,
                                                        const int32_t* __restrict__ plist, const int32_t* __restrict__ goff,
                                                        int32_t* __restrict__ gen_k, int32_t* __restrict__ gen_k2,
                                                        int32_t* __restrict__ gen_cj, const int32_t* __restrict__ pbslot) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= no) return;
  const int k = clist[q], ci = ocam[k], p = opt[k];
  if (pbslot != nullptr && pbslot[p] >= 0) return;
  int o = goff[q];
  for (int q2 = pstart[p]; q2 < pstart[p + 1]; ++q2) {
    const int k2 = plist[q2], cj = ocam[k2];
    if (cj > ci) continue;
    gen_k[o] = k;
    gen_k2[o] = k2;
    gen_cj[o] = cj;
    ++o;
  }
}

// one workgroup per camera: STABLE counting sort of its pairs by partner camera (cj <= ci: at most ci + 1 bins in LDS),
// sorted pairs to their final place, block starts and partner cameras of its blocks into the scratch arrays (indexed
// from the camera's first pair), block count to blk_cnt.  Any number of pairs per camera.
//   histogram (LDS atomics: only counts, order-free) -> exclusive scan of the bins = where each block starts ->
//   placement in generation order, 256 pairs at a time: rank among the equal keys of the chunk by a plain count over the
//   chunk (same-address LDS reads), the last of them moves the bin's cursor on.
// ranges of kPairSortRange pairs per camera (exclusive scan over the cameras, nc + 1 entries): one workgroup sorts one range
constexpr int kPairSortRange = 4096;
__global__ __launch_bounds__(256) void pair_ranges_kernel(const int32_t* __restrict__ cstart, const int32_t* __restrict__ goff, int nc,
                                                          int32_t* __restrict__ range_off) {
  __shared__ int wave_tot[4];
  int carry = 0;
  for (int c0 = 0; c0 < nc; c0 += 256) {
    const int ci = c0 + threadIdx.x;
    int v = 0;
    if (ci < nc) {
      const int m = goff[cstart[ci + 1]] - goff[cstart[ci]];
      v = m > 0 ? (m + kPairSortRange - 1) / kPairSortRange : 1;  // (an empty camera still writes its block count)
    }
    int total;
    const int ex = block_excl_scan(v, wave_tot, &total);
    if (ci < nc) range_off[ci] = carry + ex;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) range_off[nc] = carry;
}
// Stable counting sort of the pairs of camera ci by their second camera.  A workgroup takes ONE range of kPairSortRange
// pairs (a graph has cameras with tens of thousands of pairs: with a workgroup per camera the launch lasted as long as
// the biggest of them, 0.63 ms at C4 for 0.04 ms of work per CU): it counts the camera's keys (all of them: the bin
// starts), then the keys in front of its range (added to the starts: its cursors), then ranks its own pairs chunk by chunk.
__global__ __launch_bounds__(256) void pair_sort_kernel(const int32_t* __restrict__ cstart, const int32_t* __restrict__ goff,
                                                        const int32_t* __restrict__ range_off, int nc,
                                                        const int32_t* __restrict__ gen_k,
                                                        const int32_t* __restrict__ gen_k2,
                                                        const int32_t* __restrict__ gen_cj, int32_t* __restrict__ pair_a,
                                                        int32_t* __restrict__ pair_b, int32_t* __restrict__ tmp_start,
                                                        int32_t* __restrict__ tmp_cj, int32_t* __restrict__ blk_cnt) {
  extern __shared__ __attribute__((aligned(16))) int bins[];  // ci + 1 cursors
  __shared__ int wave_tot[4];
  __shared__ __attribute__((aligned(16))) int chunk_cj[256];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= range_off[nc]) return;
  int lo = 0, hi = nc - 1;  // the camera whose ranges hold this workgroup: last ci with range_off[ci] <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (range_off[mid] <= (int)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const int ci = lo, nbins = ci + 1, range = (int)blockIdx.x - range_off[ci];
  const int begin = goff[cstart[ci]], end = goff[cstart[ci + 1]], m = end - begin;
  const int r0 = range * kPairSortRange, r1 = r0 + kPairSortRange < m ? r0 + kPairSortRange : m;
  for (int i = tid; i < nbins; i += 256) bins[i] = 0;
  __syncthreads();
  for (int i = tid; i < m; i += 256) atomicAdd(&bins[gen_cj[begin + i]], 1);
  __syncthreads();
  // exclusive scan of the counts (bin -> first position) and of the non-empty flags (bin -> local block number)
  int carry = 0, bcarry = 0;
  for (int b0 = 0; b0 < nbins; b0 += 256) {
    const int bi = b0 + tid;
    const int c = bi < nbins ? bins[bi] : 0;
    int total, btotal;
    const int ex = block_excl_scan(c, wave_tot, &total);
    const int bex = block_excl_scan(c > 0 ? 1 : 0, wave_tot, &btotal);
    if (bi < nbins) {
      bins[bi] = carry + ex;
      if (c > 0 && range == 0) {
        tmp_start[begin + bcarry + bex] = begin + carry + ex;
        tmp_cj[begin + bcarry + bex] = bi;
      }
    }
    carry += total;
    bcarry += btotal;
    __syncthreads();
  }
  if (tid == 0 && range == 0) blk_cnt[ci] = bcarry;
  for (int i = tid; i < r0; i += 256) atomicAdd(&bins[gen_cj[begin + i]], 1);  // the keys in front of this range
  __syncthreads();
  for (int c0 = r0; c0 < r1; c0 += 256) {
    const int i = c0 + tid, n_here = r1 - c0 < 256 ? r1 - c0 : 256;
    const int cj = i < r1 ? gen_cj[begin + i] : -1;
    chunk_cj[tid] = cj;
    __syncthreads();
    int before = 0, all = 0;
    (void)n_here;  // the tail of the last chunk holds -1, which matches no key of a valid lane
    const int4* c4 = reinterpret_cast<const int4*>(chunk_cj);
#pragma unroll 4
    for (int t4 = 0; t4 < 64; ++t4) {  // four keys per (same-address) 16-byte LDS read
      const int4 w = c4[t4];
      const int s0 = w.x == cj, s1 = w.y == cj, s2 = w.z == cj, s3 = w.w == cj;
      all += (s0 + s1) + (s2 + s3);
      const int t = 4 * t4;
      before += (t < tid ? s0 : 0) + (t + 1 < tid ? s1 : 0) + (t + 2 < tid ? s2 : 0) + (t + 3 < tid ? s3 : 0);
    }
    if (i < r1) {
      const int dst = begin + bins[cj] + before;
      pair_a[dst] = gen_k[begin + i];
      pair_b[dst] = gen_k2[begin + i];
    }
    __syncthreads();  // every cursor of this chunk has been read
    if (i < r1 && before == all - 1) bins[cj] += all;
    __syncthreads();
  }
}

// block tables from the per-camera scratch: blk_off = exclusive scan of blk_cnt (nc + 1 entries)
__global__ __launch_bounds__(256) void pair_blocks_kernel(const int32_t* __restrict__ cstart, const int32_t* __restrict__ goff,
                                                          const int32_t* __restrict__ blk_off,
                                                          const int32_t* __restrict__ tmp_start,
                                                          const int32_t* __restrict__ tmp_cj, int nc,
                                                          int32_t* __restrict__ bstart, int32_t* __restrict__ bci,
                                                          int32_t* __restrict__ bcj, int32_t* __restrict__ seg_cnt) {
  const int ci = blockIdx.x;
  const int begin = goff[cstart[ci]], end = goff[cstart[ci + 1]];
  const int b0 = blk_off[ci], nb = blk_off[ci + 1] - b0;
  for (int lb = threadIdx.x; lb < nb; lb += 256) {
    const int st = tmp_start[begin + lb], nx = lb + 1 < nb ? tmp_start[begin + lb + 1] : end;
    bstart[b0 + lb] = st;
    bci[b0 + lb] = ci;
    bcj[b0 + lb] = tmp_cj[begin + lb];
    seg_cnt[b0 + lb] = (nx - st + kSchurSeg - 1) / kSchurSeg;
  }
  if (ci == nc - 1 && threadIdx.x == 0) bstart[blk_off[nc]] = end;  // = number of pairs
}
__global__ __launch_bounds__(256) void pair_segments_kernel(int nblocks, const int32_t* __restrict__ seg_first,
                                                            int32_t* __restrict__ seg_blk) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= nblocks) return;
  for (int sg = seg_first[b]; sg < seg_first[b + 1]; ++sg) seg_blk[sg] = b;
}

// 64 threads per block: thread t < 42 adds element t of the block's segment totals in segment order, then
// S[block] -= total (this thread is the element's only writer), rhs += total for the diagonal blocks
// (the right-hand side is also kept as row `rhs_row` of S, where the factorisation picks it up)
// what the two single-launch kernels behind the Schur product need cleared / filled before they start: done by the tail
// workgroups of schur_reduce_kernel instead of three memset nodes in the stream
struct SolveState {
  unsigned* flow_flags;  // zero
  unsigned n_flow;
  unsigned* xh_words;    // the sentinel of bwd_chain_kernel in every 32-bit word
  unsigned n_xh;
  int* info;             // zero
};
__global__ __launch_bounds__(256) void schur_reduce_kernel(SchurBlocks B, const double* __restrict__ partial,
                                                           double* __restrict__ S, int n, double* __restrict__ rhs,
                                                           int rhs_row, int n_reduce_blocks, SolveState st, CrMap map) {
  if ((int)blockIdx.x >= n_reduce_blocks) {
    const unsigned i = ((unsigned)blockIdx.x - (unsigned)n_reduce_blocks) * 256u + threadIdx.x;
    if (i < st.n_flow) st.flow_flags[i] = 0u;
    if (i < st.n_xh) st.xh_words[i] = 0xFFF8BEEFu;
    if (i == 0 && st.info) *st.info = 0;
    return;
  }
  const int blk = blockIdx.x * 4 + (threadIdx.x >> 6), t = threadIdx.x & 63;
  if (blk >= B.nblocks || t >= 42) return;
  double tot = 0;
  for (int sg = B.seg_first[blk]; sg < B.seg_first[blk + 1]; ++sg) tot += partial[(size_t)42 * sg + t];
  const int ci = B.bci[blk], cj = B.bcj[blk];
  if (t < 36) {
    const int a = t / 6, b = t - 6 * a;
    if (ci == cj && a < b) return;  // lower triangle only
    double* dst = s_elem(S, n, map, 6 * ci + a, 6 * cj + b);
    *dst = *dst - tot;
  } else if (ci == cj) {
    const double v = rhs[6 * ci + (t - 36)] + tot;
    rhs[6 * ci + (t - 36)] = v;
    *s_elem(S, n, map, rhs_row, 6 * ci + (t - 36)) = v;
  }
}

// E of a point border: for every observation k of border point p (slot b) by camera c, the 3 x 6 block W_k^T goes to rows
// n_band + 3 b .. + 2, columns 6 c .. + 5 (W_k = J_c^T J_p, 6 x 3 row-major in Wbuf).  One workgroup per border point, thread
// (a, j) walks the point's observations in list order (a camera that saw the point twice adds twice: fixed order, no atomics).
__global__ __launch_bounds__(64) void border_points_kernel(Problem P, PointBorder pb, const double* __restrict__ Wbuf,
                                                           double* __restrict__ S, int lda, int n_band, CrMap map) {
  const int b = blockIdx.x, t = threadIdx.x;
  if (b >= pb.n_pts || t >= 18) return;
  const int a = t / 3, j = t - 3 * a, p = pb.pts[b];
  for (int q = P.pstart[p]; q < P.pstart[p + 1]; ++q) {
    const int k = P.plist[q], c = P.ocam[k];
    double* dst = S + (size_t)(6 * c + a) * lda + (n_band + 3 * b + j) + map.bshift();
    *dst += Wbuf[(size_t)18 * k + 3 * a + j];
  }
}

// rhs -> row n of S (rides through the factorisation, becomes y = L^-1 rhs; the back-substitution reads it there)
__global__ void rhs_to_row_kernel(const double* __restrict__ rhs, double* __restrict__ S, int lda, int n, CrMap map) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) S[(size_t)j * lda + n + map.bshift()] = rhs[j];
}

// back-substitution for the points and the candidate state in one launch: thread i handles point i and camera i; it also
// clears the two flag words the NEXT iteration accumulates into (their read-back copies are already queued ahead of it)
__global__ __launch_bounds__(256) void backsub_update_kernel(Problem P, const double* __restrict__ Hpi,
                                                             const double* __restrict__ gp,
                                                             const double* __restrict__ dc, double* __restrict__ dp,
                                                             const double* __restrict__ Wbuf,
                                                             double* __restrict__ poses_new, double* __restrict__ pts_new,
                                                             int* __restrict__ bad, unsigned long long* __restrict__ gmax,
                                                             unsigned long long* __restrict__ keep,
                                                             const int32_t* __restrict__ pbslot, int n_band) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) {
    if (keep) {  // (what the iteration's publication sends to the host: see reduce_publish_kernel)
      keep[0] = *gmax;
      keep[1] = (unsigned long long)(unsigned)*bad;
    }
    *bad = 0;
    *gmax = 0ull;
  }
  if (i < P.nc) {
    if ((P.dof[i] & 63) == 0) {  // fixed keyframe: bitwise untouched
      for (int a = 0; a < 7; ++a) poses_new[7 * i + a] = P.poses[7 * i + a];
    } else {
      se3_retract(P.poses + 7 * i, dc + 6 * i, poses_new + 7 * i);
    }
  }
  if (i >= P.np) return;
  const int p = i;
  double rhs[3] = {-gp[3 * p], -gp[3 * p + 1], -gp[3 * p + 2]};
  for (int q = P.pstart[p]; q < P.pstart[p + 1]; ++q) {
    const int k = P.plist[q], ci = P.ocam[k];
    double W[18];
    load_W(Wbuf, k, W);
#pragma unroll
    for (int b = 0; b < 3; ++b) {  // rhs -= W^T dc
      double t = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) t += W[3 * a + b] * dc[6 * ci + a];
      rhs[b] -= t;
    }
  }
  const double* Hi = Hpi + (size_t)9 * p;
  const int bslot = pbslot != nullptr ? pbslot[p] : -1;  // a border point: its step came out of the linear solve with the cameras'
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    double d = Hi[3 * a] * rhs[0] + Hi[3 * a + 1] * rhs[1] + Hi[3 * a + 2] * rhs[2];
    if (bslot >= 0) d = dc[n_band + 3 * bslot + a];
    dp[3 * p + a] = d;
    pts_new[3 * p + a] = P.pts[3 * p + a] + d;
  }
}

// per-observation: robust cost at `Pn` state and (optionally) model decrease at the `P` linearisation.
// Block partial sums in fixed order -> partial[2 * block + {0,1}].
__global__ __launch_bounds__(256) void eval_kernel(Problem P, const double* __restrict__ poses_eval,
                                                   const double* __restrict__ pts_eval, const double* __restrict__ dc,
                                                   const double* __restrict__ dp, int with_model,
                                                   double* __restrict__ partial) {
  __shared__ double sc[256], sm[256];
  const int k = blockIdx.x * 256 + threadIdx.x;
  double cost = 0, model = 0;
  if (k < P.no) {
    const int ci = P.ocam[k], pi = P.opt[k];
    const double* info = P.oinfo ? P.oinfo + 4 * k : nullptr;
    Obs o;
    const bool in_front = linearize<false>(poses_eval + 7 * ci, 0, pts_eval + 3 * pi, 0, P.oxy + 2 * k, info, P.huber, o);
    if (in_front) cost = (P.huber > 0 && o.s > P.huber * P.huber) ? 2.0 * P.huber * sqrt(o.s) - P.huber * P.huber : o.s;
    const bool was_in_front = with_model && lin_obs(P, k, o, true);
    // a candidate that moves a valid observation behind its camera would drop it from the sum and LOWER the cost:
    // infinite cost instead, so the LM loop rejects the step (same rule in oracle/ba_oracle.c)
    if (was_in_front && !in_front) cost = __builtin_inf();
    if (was_in_front) {
      double L[4];
      weighted_info(info, o.w, L);
      double Jd[2] = {0, 0};
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        Jd[0] += o.Jc[a] * dc[6 * ci + a];
        Jd[1] += o.Jc[6 + a] * dc[6 * ci + a];
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        Jd[0] += o.Jp[a] * dp[3 * pi + a];
        Jd[1] += o.Jp[3 + a] * dp[3 * pi + a];
      }
      const double LJd[2] = {L[0] * Jd[0] + L[1] * Jd[1], L[2] * Jd[0] + L[3] * Jd[1]};
      const double Lr[2] = {L[0] * o.r[0] + L[1] * o.r[1], L[2] * o.r[0] + L[3] * o.r[1]};
      model = -(Jd[0] * Lr[0] + Jd[1] * Lr[1] + 0.5 * (Jd[0] * LJd[0] + Jd[1] * LJd[1]));
    }
  }
  sc[threadIdx.x] = cost;
  sm[threadIdx.x] = model;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (threadIdx.x < off) {
      sc[threadIdx.x] += sc[threadIdx.x + off];
      sm[threadIdx.x] += sm[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = sc[0];
    partial[2 * blockIdx.x + 1] = sm[0];
  }
}

__global__ __launch_bounds__(256) void reduce_final_kernel(const double* __restrict__ partial, int nblocks,
                                                           double* __restrict__ out) {
  __shared__ double sc[256], sm[256];
  double c = 0, m = 0;
  for (int i = threadIdx.x; i < nblocks; i += 256) {
    c += partial[2 * i];
    m += partial[2 * i + 1];
  }
  sc[threadIdx.x] = c;
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (threadIdx.x < off) {
      sc[threadIdx.x] += sc[threadIdx.x + off];
      sm[threadIdx.x] += sm[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = 0.5 * sc[0];
    out[1] = sm[0];
  }
}

// The same reduction, and everything the host decides an LM iteration on -- candidate cost, model decrease, gradient
// maximum, damping and factorisation flags -- written STRAIGHT into pinned host memory, the stamp last (system-scope
// release): the host polls the stamp instead of waiting for four small device-to-host copies and the stream (each copy is
// a launch of its own on the stream: ~6 us apiece, and the stream may then go on with the next linearisation at once).
struct HostVerdict {
  double cost, model;
  unsigned long long gmax_bits;
  int info, bad;
  int32_t pair_counts[4];
  volatile unsigned stamp;
};
__global__ __launch_bounds__(256) void reduce_publish_kernel(const double* __restrict__ partial, int nblocks,
                                                             const unsigned long long* __restrict__ keep,
                                                             const int* __restrict__ info, HostVerdict* host, unsigned stamp) {
  __shared__ double sc[256], sm[256];
  double c = 0, m = 0;
  for (int i = threadIdx.x; i < nblocks; i += 256) {
    c += partial[2 * i];
    m += partial[2 * i + 1];
  }
  sc[threadIdx.x] = c;
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (threadIdx.x < off) {
      sc[threadIdx.x] += sc[threadIdx.x + off];
      sm[threadIdx.x] += sm[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    host->cost = 0.5 * sc[0];
    host->model = sm[0];
    host->gmax_bits = keep[0];
    host->bad = (int)keep[1];
    host->info = *info;
    __threadfence_system();
    __hip_atomic_store(const_cast<unsigned*>(&host->stamp), stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Device buffers of one solve, bump-allocated from the context's grow-only arena (ctx->ba_arena); anything that does
// not fit falls back to its own hipMalloc and is freed when the solve ends.
// Sub-allocator over ONE device arena: the context's grow-only arena (one-shot gh_ba_solve calls reuse it from solve to
// solve) or an arena owned by a gh_ba_graph (resident across solves, untouched by other solves on the context).
struct DevBuf {
  gh_ctx* ctx;
  void** arena;         // -> ctx->ba_arena or the graph's own
  size_t* arena_bytes;
  std::vector<void*> ptrs;
  size_t used = 0;
  explicit DevBuf(gh_ctx* c) : ctx(c), arena(&c->ba_arena), arena_bytes(&c->ba_arena_bytes) {}
  DevBuf(gh_ctx* c, void** a, size_t* ab) : ctx(c), arena(a), arena_bytes(ab) {}
  ~DevBuf() {
    hipStreamSynchronize(ctx->stream);
    for (void* p : ptrs) hipFree(p);
  }
  gh_status reserve(size_t bytes) {
    if (bytes <= *arena_bytes) return GH_OK;
    GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (*arena) GH_HIP(ctx, hipFree(*arena));
    *arena = nullptr;
    *arena_bytes = 0;
    const size_t want = bytes + bytes / 8 + (1 << 20);
    if (hipMalloc(arena, want) != hipSuccess) {
      *arena = nullptr;
      return GH_OK;  // every buffer will take the individual fallback path
    }
    *arena_bytes = want;
    return GH_OK;
  }
  template <typename T>
  gh_status alloc(T** out, size_t count) {
    const size_t bytes = (((count ? count : 1) * sizeof(T)) + 255) & ~(size_t)255;
    if (*arena && used + bytes <= *arena_bytes) {
      *out = (T*)((char*)*arena + used);
      used += bytes;
      return GH_OK;
    }
    void* p = nullptr;
    gh_status s = gh_dev_alloc(ctx, bytes, &p);
    if (s == GH_OK) {
      ptrs.push_back(p);
      *out = (T*)p;
    }
    return s;
  }
  template <typename T>
  gh_status upload(T** out, const T* src, size_t count) {
    GH_TRY(alloc(out, count));
    if (count) GH_HIP(ctx, hipMemcpyAsync(*out, src, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return GH_OK;
  }
};

void build_csr(const int32_t* key, int n_items, int n_keys, std::vector<int32_t>& start, std::vector<int32_t>& list) {
  start.assign((size_t)n_keys + 1, 0);
  list.resize((size_t)(n_items > 0 ? n_items : 1));
  for (int k = 0; k < n_items; ++k) start[key[k] + 1]++;
  for (int i = 0; i < n_keys; ++i) start[i + 1] += start[i];
  std::vector<int32_t> fill(start.begin(), start.end() - 1);
  for (int k = 0; k < n_items; ++k) list[fill[key[k]]++] = k;
}

// The same lists by T cooperating tasks (all T must be running at the same time: they meet at two spin barriers).  Task t
// counts the keys of its slice of the items, the key ranges are prefixed in parallel (start[] and, per task, the first
// position of each key for that task's slice), then every task fills its slice in item order: stable, element for element
// what build_csr gives.
struct CsrShared {
  std::vector<int32_t> counts;  // [T][n_keys]
  std::vector<long long> range_sum;
  std::atomic<int> arrived[2];
  void reset(int T, int n_keys) {
    counts.assign((size_t)T * n_keys, 0);
    range_sum.assign((size_t)T, 0);
    arrived[0] = arrived[1] = 0;
  }
};
void build_csr_task(const int32_t* key, int n_items, int n_keys, std::vector<int32_t>& start, std::vector<int32_t>& list, int T,
                    int t, CsrShared& sh) {
  auto barrier = [&](int which) {
    sh.arrived[which].fetch_add(1, std::memory_order_acq_rel);
    while (sh.arrived[which].load(std::memory_order_acquire) < T) std::this_thread::yield();
  };
  const int i0 = (int)((long long)n_items * t / T), i1 = (int)((long long)n_items * (t + 1) / T);
  int32_t* mine = sh.counts.data() + (size_t)t * n_keys;
  for (int k = i0; k < i1; ++k) mine[key[k]]++;
  barrier(0);
  const int k0 = (int)((long long)n_keys * t / T), k1 = (int)((long long)n_keys * (t + 1) / T);
  long long sum = 0;
  for (int kk = k0; kk < k1; ++kk)
    for (int tt = 0; tt < T; ++tt) sum += sh.counts[(size_t)tt * n_keys + kk];
  sh.range_sum[t] = sum;
  barrier(1);
  long long pos = 0;
  for (int tt = 0; tt < t; ++tt) pos += sh.range_sum[tt];
  for (int kk = k0; kk < k1; ++kk) {
    start[kk] = (int32_t)pos;
    for (int tt = 0; tt < T; ++tt) {
      int32_t& c = sh.counts[(size_t)tt * n_keys + kk];
      const int32_t n = c;
      c = (int32_t)pos;  // the slot now holds task tt's first position for this key
      pos += n;
    }
  }
  if (t == T - 1) start[n_keys] = (int32_t)pos;
  // every task's offsets must be final before anybody fills: a third meeting, on the first counter (it only grows)
  sh.arrived[0].fetch_add(1, std::memory_order_acq_rel);
  while (sh.arrived[0].load(std::memory_order_acquire) < 2 * T) std::this_thread::yield();
  for (int k = i0; k < i1; ++k) list[mine[key[k]]++] = k;
}

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// Pair list of the deterministic Schur product: for every camera pair (ci >= cj) that shares a point, the list of
// observation pairs (k of ci, k2 of cj) on a common point, grouped by destination block in (ci, cj) order with the
// generation order kept inside a group.  Cameras are independent: contiguous ranges of cameras (balanced by observation
// count) are tasks for the host pool, each writes its pairs to a buffer of its own and, once every range knows where it
// starts, copies them to their final place -- the result does not depend on the number of threads or ranges.
struct PairChunk {
  std::vector<int32_t> pa, pb, bs, ci, cj;  // bs relative to the chunk
};

// pcam[q] = camera of the observation plist[q] (the inner loop then reads two contiguous arrays)
void build_pair_chunk(const gh_ba_problem* pr, int nc, int c_lo, int c_hi, const std::vector<int32_t>& pstart,
                      const std::vector<int32_t>& plist, const std::vector<int32_t>& pcam, const std::vector<int32_t>& cstart,
                      const std::vector<int32_t>& clist, PairChunk& out, const int32_t* pbslot) {
  std::vector<int32_t> cnt((size_t)nc, 0), off((size_t)nc, 0), touched, tk, tk2, tc;
  size_t total = 0;
  for (int ci = c_lo; ci < c_hi; ++ci) total += (size_t)(cstart[ci + 1] - cstart[ci]);
  out.pa.reserve(total * 4);
  out.pb.reserve(total * 4);
  for (int ci = c_lo; ci < c_hi; ++ci) {
    // pass A: the pairs of this camera in generation order (k ascending, then k2 ascending), counted per partner
    touched.clear();
    tk.clear();
    tk2.clear();
    tc.clear();
    for (int q = cstart[ci]; q < cstart[ci + 1]; ++q) {
      const int k = clist[q], p = pr->obs_point[k];
      if (pbslot != nullptr && pbslot[p] >= 0) continue;  // (a border point: no pairs)
      for (int q2 = pstart[p]; q2 < pstart[p + 1]; ++q2) {
        const int cj = pcam[q2];
        if (cj > ci) continue;
        if (cnt[cj]++ == 0) touched.push_back(cj);
        tk.push_back(k);
        tk2.push_back(plist[q2]);
        tc.push_back(cj);
      }
    }
    std::sort(touched.begin(), touched.end());
    size_t run = out.pa.size();
    for (int cj : touched) {
      out.bs.push_back((int32_t)run);
      out.ci.push_back(ci);
      out.cj.push_back(cj);
      off[cj] = (int32_t)run;
      run += (size_t)cnt[cj];
    }
    out.pa.resize(run);
    out.pb.resize(run);
    // pass B: stable scatter by partner camera (from the compact per-camera list: no second walk over the graph)
    for (size_t e = 0; e < tk.size(); ++e) {
      const int32_t pos = off[tc[e]]++;
      out.pa[pos] = tk[e];
      out.pb[pos] = tk2[e];
    }
    for (int cj : touched) cnt[cj] = 0;
  }
}

void build_schur_pairs(const gh_ba_problem* pr, int nc, const std::vector<int32_t>& pstart,
                       const std::vector<int32_t>& plist, const std::vector<int32_t>& cstart,
                       const std::vector<int32_t>& clist, std::vector<int32_t>& pair_a, std::vector<int32_t>& pair_b,
                       std::vector<int32_t>& bstart, std::vector<int32_t>& bci, std::vector<int32_t>& bcj,
                       const int32_t* pbslot) {
  const int no = pr->n_obs;
  const bool timing = getenv("GSLAM_HIP_BA_TIMING") != nullptr;
  const double t0 = now_ms();
  HostPool& pool = HostPool::get();
  int nchunk = no >= 20000 ? 2 * pool.size() : 1;  // two ranges per thread: the pair count per observation varies
  if (nchunk > nc) nchunk = nc;
  std::vector<int32_t> pcam((size_t)(no > 0 ? no : 1));
  for (int q = 0; q < no; ++q) pcam[q] = pr->obs_cam[plist[q]];
  std::vector<PairChunk> chunks((size_t)nchunk);
  std::vector<int> bound((size_t)nchunk + 1, nc);
  bound[0] = 0;
  for (int t = 1, c = 0; t < nchunk; ++t) {  // camera ranges with ~equal observation counts
    const long long target = (long long)no * t / nchunk;
    while (c < nc && cstart[c] < target) ++c;
    bound[t] = c;
  }
  const double t1 = now_ms();
  pool.run(nchunk, [&](int t) { build_pair_chunk(pr, nc, bound[t], bound[t + 1], pstart, plist, pcam, cstart, clist, chunks[t], pbslot); });
  const double t2 = now_ms();
  std::vector<size_t> po((size_t)nchunk + 1, 0), bo((size_t)nchunk + 1, 0);
  for (int t = 0; t < nchunk; ++t) {
    po[t + 1] = po[t] + chunks[t].pa.size();
    bo[t + 1] = bo[t] + chunks[t].bs.size();
  }
  const size_t np_total = po[nchunk], nb_total = bo[nchunk];
  pair_a.resize(np_total);
  pair_b.resize(np_total);
  bstart.resize(nb_total + 1);
  bci.resize(nb_total);
  bcj.resize(nb_total);
  const double t3 = now_ms();
  pool.run(nchunk, [&](int t) {
    const PairChunk& c = chunks[t];
    if (!c.pa.empty()) {
      memcpy(pair_a.data() + po[t], c.pa.data(), c.pa.size() * sizeof(int32_t));
      memcpy(pair_b.data() + po[t], c.pb.data(), c.pb.size() * sizeof(int32_t));
    }
    for (size_t i = 0; i < c.bs.size(); ++i) {
      bstart[bo[t] + i] = (int32_t)(po[t] + (size_t)c.bs[i]);
      bci[bo[t] + i] = c.ci[i];
      bcj[bo[t] + i] = c.cj[i];
    }
  });
  bstart[nb_total] = (int32_t)np_total;
  if (timing)
    fprintf(stderr, "[gh_ba] pair lists: pcam %.2f, %d ranges on %d threads %.2f, allocate %.2f, gather %.2f ms\n", t1 - t0, nchunk,
            pool.size(), t2 - t1, t3 - t2, now_ms() - t3);
}

// Device build of the pair lists (kernels above).  `pairs_ub` >= the number of pairs.  On return SB points at the device
// tables, *counts = {pairs, blocks, segments, 0} is a pinned host block that is valid after the NEXT stream
// synchronisation (the caller's first cost evaluation): nothing here waits.
gh_status build_schur_pairs_device(gh_ctx* ctx, DevBuf& db, int nc, int no, size_t pairs_ub, size_t blocks_ub,
                                   const int32_t* d_ocam, const int32_t* d_opt, const int32_t* d_pstart,
                                   const int32_t* d_plist, const int32_t* d_cstart, const int32_t* d_clist,
                                   SchurBlocks* SB, int32_t* counts_pinned, const int32_t* d_pbslot) {
  const size_t segs_ub = pairs_ub / kSchurSeg + blocks_ub + 2;
  int32_t *d_goff, *d_sums, *d_gen_k, *d_gen_k2, *d_gen_cj, *d_pa, *d_pb, *d_tmp_start, *d_tmp_cj, *d_blk_off, *d_bs, *d_bci,
      *d_bcj, *d_seg_first, *d_seg_blk, *d_sums2, *d_counts;
  const int chunks = gh_div_up(no, 1024), bchunks = gh_div_up((long long)blocks_ub + 1, 1024);
  GH_TRY(db.alloc(&d_goff, (size_t)no + 1));
  GH_TRY(db.alloc(&d_sums, (size_t)chunks + 1));
  GH_TRY(db.alloc(&d_gen_k, pairs_ub));
  GH_TRY(db.alloc(&d_gen_k2, pairs_ub));
  GH_TRY(db.alloc(&d_gen_cj, pairs_ub));
  GH_TRY(db.alloc(&d_pa, pairs_ub));
  GH_TRY(db.alloc(&d_pb, pairs_ub));
  GH_TRY(db.alloc(&d_tmp_start, pairs_ub));
  GH_TRY(db.alloc(&d_tmp_cj, pairs_ub));
  GH_TRY(db.alloc(&d_blk_off, (size_t)nc + 2));
  int32_t* d_range_off;
  GH_TRY(db.alloc(&d_range_off, (size_t)nc + 2));
  GH_TRY(db.alloc(&d_bs, blocks_ub + 1));
  GH_TRY(db.alloc(&d_bci, blocks_ub + 1));
  GH_TRY(db.alloc(&d_bcj, blocks_ub + 1));
  GH_TRY(db.alloc(&d_seg_first, blocks_ub + 2));
  GH_TRY(db.alloc(&d_seg_blk, segs_ub));
  GH_TRY(db.alloc(&d_sums2, (size_t)bchunks + 1));
  GH_TRY(db.alloc(&d_counts, 4));
  GH_HIP(ctx, hipMemsetAsync(d_counts, 0, 4 * sizeof(int32_t), ctx->stream));
  GH_HIP(ctx, hipMemsetAsync(d_seg_first, 0, (blocks_ub + 2) * sizeof(int32_t), ctx->stream));  // counts beyond nblocks: 0
  const dim3 T(256);
  // pairs per observation (camera-major order), exclusive scan -> goff (no + 1 entries), total -> counts[0]
  GH_LAUNCH(ctx, "ba_pl_count", pair_count_kernel, dim3(gh_div_up(no, 256)), T, 0, no, d_clist, d_ocam, d_opt, d_pstart,
            d_plist, d_goff, d_pbslot);
  GH_LAUNCH(ctx, "ba_pl_scan", scan_chunks_kernel, dim3(chunks), T, 0, (const int32_t*)d_goff, no, d_goff, d_sums);
  GH_LAUNCH(ctx, "ba_pl_scan", scan_sums_kernel, dim3(1), T, 0, d_sums, chunks, d_counts + 0);
  GH_LAUNCH(ctx, "ba_pl_scan", scan_add_kernel, dim3(gh_div_up(no, 256)), T, 0, d_goff, no, (const int32_t*)d_sums, chunks);
  GH_LAUNCH(ctx, "ba_pl_emit", pair_emit_kernel, dim3(gh_div_up(no, 256)), T, 0, no, d_clist, d_ocam, d_opt, d_pstart,
            d_plist, (const int32_t*)d_goff, d_gen_k, d_gen_k2, d_gen_cj, d_pbslot);
  static bool lds_attr[64] = {};
  const int dev = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
  if (!lds_attr[dev]) {
    GH_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(pair_sort_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kPairSortMaxCams * 4));
    lds_attr[dev] = true;
  }
  GH_LAUNCH(ctx, "ba_pl_scan", pair_ranges_kernel, dim3(1), T, 0, d_cstart, (const int32_t*)d_goff, nc, d_range_off);
  // (the grid is an upper bound on the number of ranges: the workgroups beyond the device's count return at once)
  GH_LAUNCH(ctx, "ba_pl_sort", pair_sort_kernel, dim3((unsigned)(pairs_ub / kPairSortRange + (size_t)nc)), T,
            (size_t)nc * sizeof(int), d_cstart, (const int32_t*)d_goff, (const int32_t*)d_range_off, nc, (const int32_t*)d_gen_k, (const int32_t*)d_gen_k2, (const int32_t*)d_gen_cj, d_pa, d_pb, d_tmp_start, d_tmp_cj,
            d_blk_off);
  // blocks: exclusive scan of the per-camera block counts (in place, nc + 1 entries), total -> counts[1]
  GH_LAUNCH(ctx, "ba_pl_scan", scan_sums_kernel, dim3(1), T, 0, d_blk_off, nc, d_counts + 1);
  GH_LAUNCH(ctx, "ba_pl_blocks", pair_blocks_kernel, dim3(nc), T, 0, d_cstart, (const int32_t*)d_goff,
            (const int32_t*)d_blk_off, (const int32_t*)d_tmp_start, (const int32_t*)d_tmp_cj, nc, d_bs, d_bci, d_bcj,
            d_seg_first);
  // segments: exclusive scan of the per-block segment counts over blocks_ub entries (the tail beyond nblocks is zero)
  // -> seg_first (nblocks + 1 meaningful entries), total -> counts[2]
  GH_LAUNCH(ctx, "ba_pl_scan", scan_chunks_kernel, dim3(bchunks), T, 0, (const int32_t*)d_seg_first, (int)blocks_ub + 1,
            d_seg_first, d_sums2);
  GH_LAUNCH(ctx, "ba_pl_scan", scan_sums_kernel, dim3(1), T, 0, d_sums2, bchunks, d_counts + 2);
  GH_LAUNCH(ctx, "ba_pl_scan", scan_add_kernel, dim3(gh_div_up((long long)blocks_ub + 1, 256)), T, 0, d_seg_first,
            (int)blocks_ub + 1, (const int32_t*)d_sums2, bchunks);
  GH_LAUNCH(ctx, "ba_pl_segs", pair_segments_kernel, dim3(gh_div_up((long long)blocks_ub, 256)), T, 0, (int)blocks_ub,
            (const int32_t*)d_seg_first, d_seg_blk);
  GH_HIP(ctx, hipMemcpyAsync(counts_pinned, d_counts, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  *SB = SchurBlocks{d_pa, d_pb, d_bs, d_bci, d_bcj, d_seg_blk, d_seg_first, 0, 0};
  return GH_OK;
}

}  // namespace

extern "C" void gh_ba_default_options(gh_ba_options* o) {
  if (!o) return;
  o->huber_delta = 0.01;
  o->max_iterations = 500;
  o->initial_radius = 1e4;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->min_relative_decrease = 1e-3;
  o->verbose = 0;
  o->deterministic = 1;
}

namespace {

// Everything a solve keeps on the device.  A one-shot gh_ba_solve builds one on the stack over the context's arena; a
// gh_ba_graph owns one (and its arena) so that the index lists, pair lists, chunk tables and all arrays stay in HBM
// between the windowed solves of a SLAM back end (GSLAM/core/Optimizer.h:229 is called every few keyframes on a graph
// whose topology changes far less often than its values).
struct BaSession {
  // Arrow ordering (loop closures): cameras are renumbered so that the ones long-range points tie to far-away cameras come
  // last -- the reduced camera system is then a band + a dense border (chol_cr.hip, arrowhead mode).  perm[new] = old camera
  // (empty = the caller's order); everything on the device is in the new order, the entry points translate.
  std::vector<int32_t> perm;
  int n_border = 0;  // cameras of the border
  // POINT BORDER: the long-range points that stay out of the Schur complement (ba_order.hip chose them over their far cameras); their
  // 3 unknowns each follow the cameras' in the reduced system.  Never together with a camera border.
  std::vector<int32_t> border_pts;
  int32_t *d_bpts = nullptr, *d_pbslot = nullptr;
  int span_measured = -1;  // >= 0: ba_order.hip has measured the camera span of the order in use (and checked every index): ba_run skips its pass
  int reordered = 0; // the bandwidth-reducing order of ba_order.hip was applied (perm is then non-empty even without a border)
  double* d_arrow_ws = nullptr;
  // structure of the border rows of the reduced camera system (arrowhead solver): border_cam_nz[strip * nc_band + c] != 0 iff a
  // border camera with rows in the 16-row strip sees a point band camera c sees; d_border_nz = what gh_cr_border_symbolic made of it
  std::vector<uint8_t> border_cam_nz;
  uint8_t* d_border_nz = nullptr;
  bool ready = false;
  void* arena = nullptr;  // graph-owned arena (unused by one-shot solves)
  size_t arena_bytes = 0;
  DevBuf* db = nullptr;
  int nc = 0, np = 0, no = 0, deterministic = 0;
  bool has_info = false, has_pfree = false;
  double *d_poses = nullptr, *d_pts = nullptr, *d_poses_new = nullptr, *d_pts_new = nullptr, *d_oxy = nullptr, *d_oinfo = nullptr;
  int32_t *d_dof = nullptr, *d_ocam = nullptr, *d_opt = nullptr, *d_pstart = nullptr, *d_plist = nullptr, *d_cstart = nullptr,
          *d_clist = nullptr;
  uint8_t* d_pfree = nullptr;
  CamChunks CC{nullptr, nullptr, nullptr, 0};
  SchurBlocks SB{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
  int nchunks = 0, nsegs = 0, nblocks = 0;
  bool device_pairs = false;
  double *d_Hcc = nullptr, *d_gc = nullptr, *d_Hpp = nullptr, *d_gp = nullptr, *d_Hpi = nullptr, *d_S = nullptr, *d_dc = nullptr,
         *d_dp = nullptr, *d_partial = nullptr, *d_out = nullptr, *d_work = nullptr, *d_dinv = nullptr, *d_W = nullptr,
         *d_cpart = nullptr, *d_spart = nullptr, *d_xwork = nullptr, *d_xh = nullptr;
  // second set of the linearisation's outputs: the candidate state is linearised speculatively while the host decides
  double *d_W2 = nullptr, *d_Hpp2 = nullptr, *d_gp2 = nullptr, *d_cpart2 = nullptr;
  unsigned long long* d_keep = nullptr;  // {gradient maximum, damping flag} of the iteration, saved by backsub_update before it clears them
  unsigned long long* d_gmax = nullptr;
  int *d_bad = nullptr, *d_info = nullptr;
  unsigned* d_flow = nullptr;
  size_t n_pairs = 0;
  // band solver (chol_cr.hip): tiles per superblock (0 = the dense factorisation), its inverted diagonal tiles and panels
  int cr_T = 0, cam_span = 0;
  int lda = 0;   // rows per column of d_S as allocated
  CrMap map;     // layout of d_S: dense (map.m == 0) or the band solver's compact columns
  double *d_cr_dinv = nullptr, *d_cr_W = nullptr;
  ~BaSession() {
    delete db;
    if (arena) hipFree(arena);
  }
};

// pr == nullptr (graph sessions only): the values already on the device are solved as they are.  download: write the
// result back into pr->cam_pose / pr->point_xyz.
gh_status ba_run(gh_ctx* ctx, BaSession& S, gh_ba_problem* pr, const gh_ba_options* opt_in, gh_ba_summary* sum_out,
                 bool download) {
  gh_ba_options opt;
  gh_ba_default_options(&opt);
  if (opt_in) opt = *opt_in;
  gh_ba_summary local_sum;
  gh_ba_summary* sum = sum_out ? sum_out : &local_sum;
  memset(sum, 0, sizeof(*sum));
  GH_CHECK_ARG(ctx, pr != nullptr || S.ready);
  const int nc = pr ? pr->n_cams : S.nc, np = pr ? pr->n_points : S.np, no = pr ? pr->n_obs : S.no;
  GH_CHECK_ARG(ctx, nc >= 1 && np >= 0 && no >= 0 && nc <= (1 << 24));
  if (pr) {
    GH_CHECK_ARG(ctx, pr->cam_pose && pr->cam_dof && (np == 0 || pr->point_xyz));
    GH_CHECK_ARG(ctx, no == 0 || (pr->obs_cam && pr->obs_point && pr->obs_xy));
  }
  if (S.ready) {  // a resident graph: same topology by contract, checked as far as the sizes go
    GH_CHECK_ARG(ctx, nc == S.nc && np == S.np && no == S.no && (opt.deterministic != 0) == (S.deterministic != 0));
    GH_CHECK_ARG(ctx, !pr || ((pr->obs_info != nullptr) == S.has_info && (pr->point_free != nullptr) == S.has_pfree));
  } else {
    // (the same pass finds how far apart, in camera indices, the observers of one point are: a band -> the band solver)
    const int nc_band = nc - S.n_border;  // (arrow ordering: the border cameras are the last ones and do not count)
    if (S.span_measured >= 0 && S.n_border == 0) {
      S.cam_span = S.span_measured;  // (ba_order.hip: the same pass, on the pool threads, indices checked)
    } else {
      std::vector<int32_t> cam_lo((size_t)np, INT32_MAX), cam_hi((size_t)np, -1);
      for (int k = 0; k < no; ++k) {
        const int32_t c = pr->obs_cam[k], p = pr->obs_point[k];
        GH_CHECK_ARG(ctx, c >= 0 && c < nc && p >= 0 && p < np);
        if (c >= nc_band) continue;
        if (c < cam_lo[p]) cam_lo[p] = c;
        if (c > cam_hi[p]) cam_hi[p] = c;
      }
      int span = 0;
      for (int p = 0; p < np; ++p)
        if (cam_hi[p] >= 0 && cam_hi[p] - cam_lo[p] > span) span = cam_hi[p] - cam_lo[p];
      S.cam_span = span;
    }
    S.border_cam_nz.clear();
    // (GSLAM_HIP_BA_ARROW_DENSE_BORDER = 1: never, 0: always; default: when the border block has 4 M entries or more -- at C4 + 20
    //  closures, 0.8 M, the two passes below cost what the border kernels save over a dozen iterations; at C5 + 50, 51 M, one
    //  iteration pays for them)
    const char* dense_border_env = getenv("GSLAM_HIP_BA_ARROW_DENSE_BORDER");
    const bool want_structure = dense_border_env ? dense_border_env[0] == '0' : 36LL * S.n_border * nc_band >= (4LL << 20);
    if (S.n_border > 0 && want_structure) {
      // which (border strip, band camera) blocks of the reduced system can be non-zero: a common point (two more passes)
      const int nbs = (6 * S.n_border + 15) / 16;
      std::vector<int32_t> first((size_t)np, -1), next_of;   // per point: chain of its border observers
      std::vector<int32_t> cam_of;
      std::vector<uint8_t> has((size_t)np / 8 + 1, 0);        // (a bit per point: the test of the second pass stays in cache)
      for (int k = 0; k < no; ++k) {
        const int32_t c = pr->obs_cam[k];
        if (c < nc_band) continue;
        const int32_t p = pr->obs_point[k];
        cam_of.push_back(c - nc_band);
        next_of.push_back(first[p]);
        first[p] = (int32_t)cam_of.size() - 1;
        has[(size_t)p >> 3] |= (uint8_t)(1u << (p & 7));
      }
      S.border_cam_nz.assign((size_t)nbs * nc_band, 0);
      for (int k = 0; k < no; ++k) {
        const int32_t p = pr->obs_point[k];
        if (!((has[(size_t)p >> 3] >> (p & 7)) & 1)) continue;
        const int32_t c = pr->obs_cam[k];
        if (c >= nc_band) continue;
        for (int32_t e = first[p]; e >= 0; e = next_of[e]) {
          const int r0 = 6 * cam_of[e];
          S.border_cam_nz[(size_t)(r0 / 16) * nc_band + c] = 1;
          S.border_cam_nz[(size_t)((r0 + 5) / 16) * nc_band + c] = 1;
        }
      }
    }
    // the same for a POINT border: border point b (rows 3 b .. 3 b + 2) only touches the columns of the cameras that see it
    const int nbp0 = (int)S.border_pts.size();
    const bool want_pstructure = dense_border_env ? dense_border_env[0] == '0' : 18LL * nbp0 * nc >= (4LL << 20);
    if (nbp0 > 0 && want_pstructure) {
      const int nbs = (3 * nbp0 + 15) / 16;
      std::vector<int32_t> slot_of((size_t)np, -1);
      for (int b = 0; b < nbp0; ++b) slot_of[S.border_pts[b]] = b;
      S.border_cam_nz.assign((size_t)nbs * nc, 0);
      for (int k = 0; k < no; ++k) {
        const int32_t b = slot_of[pr->obs_point[k]];
        if (b < 0) continue;
        const int32_t c = pr->obs_cam[k];
        S.border_cam_nz[(size_t)((3 * b) / 16) * nc + c] = 1;
        S.border_cam_nz[(size_t)((3 * b + 2) / 16) * nc + c] = 1;
      }
    }
  }
  GH_HIP(ctx, hipSetDevice(ctx->device));
  const double t_begin = now_ms();
  const int n_bpts = (int)S.border_pts.size();  // (point border: never together with a camera border)
  const int n = 6 * nc + 3 * n_bpts;
  const int n_band = 6 * (nc - S.n_border);  // (== n unless the cameras are in arrow order / points form a border)
  // one extra row carries the right-hand side through the factorisation; columns start on 128-byte lines
  // (dense: n + 1 rows per column; band / arrowhead solver: compact columns, decided with the solver below -- cr_map.h)
  int lda = S.ready ? S.lda : (n + 1 + 15) & ~15;

  // Everything the host reads back during an iteration (gradient maximum, factorisation / damping flags, candidate cost
  // and model decrease) lands in ONE pinned block: copies into pageable memory are staged and block the host in the middle
  // of the launch chain, which left the GPU idle while the rest of the iteration was being enqueued.
  typedef HostVerdict Readback;  // cost, model, gmax_bits, info, bad, pair_counts[4] (device-built pair lists: pairs, blocks, segments), stamp
  // The caller's arrays (8.9 MB at C4, pageable: 0.85-1.2 ms of staged copies when the driver does it) go up WHILE the host
  // builds its index lists -- when the arena of an earlier solve is there to take them (a first solve, or one that has to
  // grow the arena, uploads after the lists as before).  With enough pool threads the staging is ours: the arrays lie back
  // to back on the device, four tasks copy a quarter of that image each into pinned memory and send it with one
  // asynchronous copy.
  const bool with_info = pr ? pr->obs_info != nullptr : S.has_info;
  const size_t raw_bytes = 8 * ((size_t)nc * 7 * 2 + (size_t)np * 3 * 2 + (size_t)no * 2 + (with_info ? (size_t)no * 4 : 0)) +
                           4 * ((size_t)nc + 2 * (size_t)no) + (size_t)np + 16 * 256;
  const char* early_env = getenv("GSLAM_HIP_BA_EARLY_UPLOAD");  // "0": upload after the lists (A/B measurements)
  DevBuf& db = *S.db;
  bool early = !S.ready && *db.arena != nullptr && raw_bytes <= *db.arena_bytes && !(early_env && early_env[0] == '0');
  Readback* rb = nullptr;  // (a pinned block of its own: nothing the staging below or a later gh_pinned does can move it)
  char* stage = nullptr;
  {
    void* pp = nullptr;
    GH_TRY(gh_readback_block(ctx, sizeof(Readback), &pp));
    rb = (Readback*)pp;
    if (early) {
      GH_TRY(gh_pinned(ctx, raw_bytes, &pp));
      stage = (char*)pp;
    }
  }
  double *&d_poses = S.d_poses, *&d_pts = S.d_pts, *&d_poses_new = S.d_poses_new, *&d_pts_new = S.d_pts_new, *&d_oxy = S.d_oxy,
         *&d_oinfo = S.d_oinfo;
  int32_t *&d_dof = S.d_dof, *&d_ocam = S.d_ocam, *&d_opt = S.d_opt, *&d_pstart = S.d_pstart, *&d_plist = S.d_plist,
          *&d_cstart = S.d_cstart, *&d_clist = S.d_clist;
  uint8_t*& d_pfree = S.d_pfree;
  CamChunks& CC = S.CC;
  SchurBlocks& SB = S.SB;
  int &nchunks = S.nchunks, &nsegs = S.nsegs, &nblocks = S.nblocks;
  bool& device_pairs = S.device_pairs;
  double *&d_Hcc = S.d_Hcc, *&d_gc = S.d_gc, *&d_Hpp = S.d_Hpp, *&d_gp = S.d_gp, *&d_Hpi = S.d_Hpi, *&d_S = S.d_S, *&d_dc = S.d_dc,
         *&d_dp = S.d_dp, *&d_partial = S.d_partial, *&d_out = S.d_out, *&d_work = S.d_work, *&d_dinv = S.d_dinv, *&d_W = S.d_W,
         *&d_cpart = S.d_cpart, *&d_spart = S.d_spart, *&d_xwork = S.d_xwork, *&d_xh = S.d_xh;
  double *&d_W2 = S.d_W2, *&d_Hpp2 = S.d_Hpp2, *&d_gp2 = S.d_gp2, *&d_cpart2 = S.d_cpart2;
  unsigned long long*& d_gmax = S.d_gmax;
  int *&d_bad = S.d_bad, *&d_info = S.d_info;
  unsigned*& d_flow = S.d_flow;
  int& cr_T = S.cr_T;
  double *&d_cr_dinv = S.d_cr_dinv, *&d_cr_W = S.d_cr_W;
  const int eval_blocks = gh_div_up(no > 0 ? no : 1, 256);
  // (first run only, kept for the verbose / check paths below)
  std::vector<int32_t> pair_a, pair_b, bstart, bci, bcj;
  bool pairs_check = false;
  SchurBlocks SB_host = SB;
  double t_csr = t_begin, t_lists = t_begin;
  if (!S.ready) {  // ================================================================ set-up (once per topology)
  struct RawPiece {
    char* dst;
    const char* src;
    size_t bytes;
  };
  std::vector<RawPiece> raw_pieces;
  auto raw_arrays = [&](bool copy) -> gh_status {  // copy = false: allocate and note the pieces only
    raw_pieces.clear();
    auto put = [&](auto** out, auto* src, size_t count) -> gh_status {
      GH_TRY(db.alloc(out, count));
      if (count) raw_pieces.push_back(RawPiece{(char*)*out, (const char*)src, count * sizeof(**out)});
      if (copy && count)
        GH_HIP(ctx, hipMemcpyAsync(*out, src, count * sizeof(**out), hipMemcpyHostToDevice, ctx->stream));
      return GH_OK;
    };
    GH_TRY(put(&d_poses, pr->cam_pose, (size_t)nc * 7));
    GH_TRY(put(&d_pts, pr->point_xyz, (size_t)np * 3));
    GH_TRY(put(&d_dof, pr->cam_dof, (size_t)nc));
    if (pr->point_free) GH_TRY(put(&d_pfree, pr->point_free, (size_t)np));
    GH_TRY(put(&d_ocam, pr->obs_cam, (size_t)no));
    GH_TRY(put(&d_opt, pr->obs_point, (size_t)no));
    GH_TRY(put(&d_oxy, pr->obs_xy, (size_t)no * 2));
    if (pr->obs_info) GH_TRY(put(&d_oinfo, pr->obs_info, (size_t)no * 4));
    GH_TRY(db.alloc(&d_poses_new, (size_t)nc * 7));
    GH_TRY(db.alloc(&d_pts_new, (size_t)np * 3));
    return GH_OK;
  };
  gh_status early_status = GH_OK;
  // part u of U of the device image of the arrays: pageable -> pinned, then one asynchronous copy
  auto stage_part = [&](int u, int U) -> gh_status {
    char* base = raw_pieces.front().dst;
    const size_t total = (size_t)(raw_pieces.back().dst - base) + raw_pieces.back().bytes;
    const size_t b0 = (total * u / U) & ~(size_t)255, b1 = u + 1 == U ? total : (total * (u + 1) / U) & ~(size_t)255;
    for (const RawPiece& pc : raw_pieces) {
      const size_t p0 = (size_t)(pc.dst - base), p1 = p0 + pc.bytes;
      const size_t lo = std::max(p0, b0), hi = std::min(p1, b1);
      if (lo < hi) memcpy(stage + lo, pc.src + (lo - p0), hi - lo);
    }
    if (b1 > b0) GH_HIP(ctx, hipMemcpyAsync(base + b0, stage + b0, b1 - b0, hipMemcpyHostToDevice, ctx->stream));
    return GH_OK;
  };

  std::vector<int32_t> pstart, plist, cstart, clist;
  {
    // index lists by point and by camera: each by a team of pool tasks when the pool has a thread for every task of the
    // region (the teams meet at spin barriers), else one task per list
    HostPool& pool = HostPool::get();
    constexpr int Tp = 6, Tc = 3, U = 4;
    const bool teams = no >= 20000 && pool.size() >= Tp + Tc + (early ? U : 0);
    gh_status part_status[U] = {GH_OK, GH_OK, GH_OK, GH_OK};
    if (early && teams) {
      GH_TRY(raw_arrays(false));
      early = !raw_pieces.empty() && raw_pieces.front().dst >= (char*)*db.arena &&
              raw_pieces.back().dst + raw_pieces.back().bytes <= (char*)*db.arena + *db.arena_bytes;
      if (!early) db.used = 0;  // (cannot happen with an arena that holds raw_bytes; the plain path allocates again)
    }
    const int extra = early ? (teams ? U : 1) : 0;
    auto upload_task = [&](int u) {
      (void)hipSetDevice(ctx->device);  // the current device is a per-thread setting
      if (teams) part_status[u] = stage_part(u, U);
      else early_status = raw_arrays(true);
    };
    // Large graphs: the two lists come from a stable radix sort on the GPU (csr_sort.hip) while the pool stages the raw
    // arrays -- the host teams need ~60 ms for the 6 M observations of C5.  GSLAM_HIP_BA_CSR=host keeps the host lists.
    const char* csr_env = getenv("GSLAM_HIP_BA_CSR");
    const bool gpu_csr = no >= (1 << 20) && !(csr_env && csr_env[0] == 'h');
    if (gpu_csr) {
      pstart.assign((size_t)np + 1, 0);
      cstart.assign((size_t)nc + 1, 0);
      plist.resize((size_t)no);
      clist.resize((size_t)no);
      gh_status csr_status[2] = {GH_OK, GH_OK};
      pool.run(2 + extra, [&](int t) {  // (the two sorts share the stream; their pageable host copies overlap)
        (void)hipSetDevice(ctx->device);
        if (t == 0) csr_status[0] = gh_csr_build_dev(ctx, pr->obs_point, no, np, pstart.data(), plist.data());
        else if (t == 1) csr_status[1] = gh_csr_build_dev(ctx, pr->obs_cam, no, nc, cstart.data(), clist.data());
        else upload_task(t - 2);
      });
      GH_TRY(csr_status[0]);
      GH_TRY(csr_status[1]);
      for (int u = 0; u < U; ++u) GH_TRY(part_status[u]);
    } else if (teams) {
      CsrShared shp, shc;
      shp.reset(Tp, np);
      shc.reset(Tc, nc);
      pstart.assign((size_t)np + 1, 0);
      cstart.assign((size_t)nc + 1, 0);
      plist.resize((size_t)(no > 0 ? no : 1));
      clist.resize((size_t)(no > 0 ? no : 1));
      pool.run(Tp + Tc + extra, [&](int t) {
        if (t < Tp) build_csr_task(pr->obs_point, no, np, pstart, plist, Tp, t, shp);
        else if (t < Tp + Tc) build_csr_task(pr->obs_cam, no, nc, cstart, clist, Tc, t - Tp, shc);
        else upload_task(t - Tp - Tc);
      });
      for (int u = 0; u < U; ++u) GH_TRY(part_status[u]);
    } else {
      pool.run(2 + extra, [&](int t) {
        if (t == 0) build_csr(pr->obs_point, no, np, pstart, plist);
        else if (t == 1) build_csr(pr->obs_cam, no, nc, cstart, clist);
        else upload_task(0);
      });
    }
  }
  GH_TRY(early_status);
  if (const char* pe = getenv("GSLAM_HIP_BA_PAIRS"); pe && pe[0] == 'c') {  // check mode: the team-built lists against the serial ones
    std::vector<int32_t> s0, l0, s1, l1;
    build_csr(pr->obs_point, no, np, s0, l0);
    build_csr(pr->obs_cam, no, nc, s1, l1);
    if (s0 != pstart || s1 != cstart || !std::equal(plist.begin(), plist.begin() + no, l0.begin()) ||
        !std::equal(clist.begin(), clist.begin() + no, l1.begin()))
      return gh_set_error(ctx, GH_ERR_NUMERIC, "index lists built by the pool teams differ from the serial lists");
  }
  t_csr = now_ms();

  // deterministic Schur: pair list sorted by destination block (built once; structure is iteration-invariant).  Large
  // graphs build it on the GPU (build_schur_pairs_device); GSLAM_HIP_BA_PAIRS=host keeps the host build, =check runs both
  // and compares them element for element (tests).
  const char* pairs_env = getenv("GSLAM_HIP_BA_PAIRS");
  const bool want_pairs = opt.deterministic && no > 0;
  pairs_check = want_pairs && pairs_env && pairs_env[0] == 'c';
  device_pairs = want_pairs && (no >= 20000 || pairs_check) && !(pairs_env && pairs_env[0] == 'h');
  size_t pairs_ub = 0, blocks_ub = 0;
  if (device_pairs) {
    for (int p = 0; p < np; ++p) pairs_ub += (size_t)(pstart[p + 1] - pstart[p]) * (size_t)(pstart[p + 1] - pstart[p]);
    blocks_ub = std::min(pairs_ub, (size_t)nc * ((size_t)nc + 1) / 2);
    // (a camera shares points only with the cameras at most `span` positions before it -- and with the border cameras: without
    //  this bound the segment tables of a 2 M-point graph reserved 24 GB for 0.5 M blocks)
    blocks_ub = std::min(blocks_ub, (size_t)nc * ((size_t)S.cam_span + 1) + (size_t)S.n_border * (size_t)nc);
    if (pairs_ub > (size_t)1 << 30 || nc > kPairSortMaxCams) device_pairs = false;  // 32-bit offsets, LDS cursors
  }
  std::vector<int32_t> pbslot_host;  // point border: slot of every point (-1: an ordinary point, eliminated by the Schur complement)
  if (n_bpts > 0) {
    pbslot_host.assign((size_t)np, -1);
    for (int b = 0; b < n_bpts; ++b) pbslot_host[S.border_pts[b]] = b;
  }
  if (want_pairs && (!device_pairs || pairs_check))
    build_schur_pairs(pr, nc, pstart, plist, cstart, clist, pair_a, pair_b, bstart, bci, bcj, n_bpts > 0 ? pbslot_host.data() : nullptr);
  nblocks = (int)bci.size();
  t_lists = now_ms();

  // the linear solver this session will use (the same rule again further down, where its buffers are carved): the arena is sized
  // for the band solver's COMPACT columns + workspaces when that is the one, for the dense lower triangle otherwise
  auto planned_tiles = [&]() {
    int want = ctx->ba_solver;
    if (const char* e = getenv("GSLAM_HIP_BA_SOLVER")) want = e[0] == 'd' ? 1 : (e[0] == 'b' ? 2 : 0);
    return want == 1 ? 0 : gh_cr_tiles(n_band, 6 * S.cam_span + 5);
  };
  {
    const size_t N = (size_t)n, NP = (size_t)np, NO = (size_t)no, NC = (size_t)nc;
    const int T_plan = planned_tiles();
    size_t s_doubles = N * (N + 1) + 64;
    if (T_plan) {
      const char* de = getenv("GSLAM_HIP_BA_DENSE_S");
      if (!(de && de[0] == '1')) s_doubles = N * (size_t)gh_cr_compact_lda(n_band, T_plan, n - n_band, nullptr);
      s_doubles += gh_cr_dinv_doubles(n_band, T_plan) + gh_cr_panel_doubles(n_band, T_plan) +
                   gh_arrow_ws_doubles(ctx, n_band, T_plan, n - n_band) + gh_cr_border_symbolic_bytes(n_band, T_plan, n - n_band) / 8 + 4 * 64;
    }
    const size_t need = 8 * (2 * NC * 7 + 2 * NP * 3 + NO * 2 + (pr->obs_info ? NO * 4 : 0) + NC * 36 + N * 3 + NP * 9 * 2 +
                             NP * 3 * 2 + s_doubles + (N + 64) * 64 * 3 + NO * 18 + (NO / 256 + 2) * 2 + 8) +
                        4 * (NC + NO * 4 + NP + NC + 4 + pair_a.size() * 2 + bstart.size() * 3) + NP + 64 * 256 +
                        (n_bpts > 0 ? 4 * (NP + (size_t)n_bpts) + 1024 : 0) +
                        // device-built pair lists: 7 arrays of pairs_ub ints, block / segment tables, scan scratch
                        4 * (pairs_ub * 7 + blocks_ub * 5 + pairs_ub / kSchurSeg + NO + NO / 1024 + blocks_ub / 1024 + 2 * NC + 128) +
                        (device_pairs ? 26 * 256 : 0) +
                        // chunk / segment tables and their partial sums (upper bounds)
                        (NO / kCamChunk + NC + 2) * (27 * 8 + 2 * 4) + (NC + 2) * 4 +
                        (std::max(pair_a.size(), pairs_ub) / kSchurSeg + std::max(bstart.size(), blocks_ub) + 2) * (42 * 8 + 4) +
                        (bstart.size() + 2) * 4 + 16 * 256;
    if (early && need > *db.arena_bytes) {  // the arena has to grow: what went up early goes up again
      early = false;
      db.used = 0;
    }
    GH_TRY(db.reserve(need));
  }
  if (!early) GH_TRY(raw_arrays(true));
  const double t_u0 = now_ms();
  GH_TRY(db.upload(&d_pstart, (const int32_t*)pstart.data(), pstart.size()));
  GH_TRY(db.upload(&d_plist, (const int32_t*)plist.data(), plist.size()));
  GH_TRY(db.upload(&d_cstart, (const int32_t*)cstart.data(), cstart.size()));
  GH_TRY(db.upload(&d_clist, (const int32_t*)clist.data(), clist.size()));
  const double t_u1 = now_ms();
  // camera chunks (lin_cams) and Schur segments: fixed-size work items, independent of how skewed the graph is
  std::vector<int32_t> ch_cam, ch_q0, ch_first((size_t)nc + 1, 0), seg_blk, seg_first((size_t)nblocks + 1, 0);
  for (int c = 0; c < nc; ++c) {
    ch_first[c] = (int32_t)ch_cam.size();
    for (int q = cstart[c]; q < cstart[c + 1]; q += kCamChunk) {
      ch_cam.push_back(c);
      ch_q0.push_back(q);
    }
  }
  ch_first[nc] = (int32_t)ch_cam.size();
  for (int b = 0; b < nblocks; ++b) {
    seg_first[b] = (int32_t)seg_blk.size();
    for (int e = bstart[b]; e < bstart[b + 1]; e += kSchurSeg) seg_blk.push_back(b);
  }
  seg_first[nblocks] = (int32_t)seg_blk.size();
  nchunks = (int)ch_cam.size();
  nsegs = (int)seg_blk.size();
  CC = CamChunks{nullptr, nullptr, nullptr, nchunks};
  {
    int32_t *d_cc, *d_cq, *d_cf;
    GH_TRY(db.upload(&d_cc, (const int32_t*)ch_cam.data(), ch_cam.size()));
    GH_TRY(db.upload(&d_cq, (const int32_t*)ch_q0.data(), ch_q0.size()));
    GH_TRY(db.upload(&d_cf, (const int32_t*)ch_first.data(), ch_first.size()));
    CC = CamChunks{d_cc, d_cq, d_cf, nchunks};
  }
  const double t_u2 = now_ms();
  SB = SchurBlocks{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nblocks, nsegs};
  SB_host = SB;  // (check mode: the host-built tables beside the device-built ones)
  S.d_pbslot = S.d_bpts = nullptr;
  if (n_bpts > 0) {
    GH_TRY(db.upload(&S.d_pbslot, (const int32_t*)pbslot_host.data(), pbslot_host.size()));
    GH_TRY(db.upload(&S.d_bpts, (const int32_t*)S.border_pts.data(), S.border_pts.size()));
  }
  if (device_pairs) {
    const double tq0 = now_ms();
    GH_TRY(build_schur_pairs_device(ctx, db, nc, no, pairs_ub, blocks_ub, d_ocam, d_opt, d_pstart, d_plist, d_cstart, d_clist,
                                    &SB, rb->pair_counts, S.d_pbslot));
    if (getenv("GSLAM_HIP_BA_TIMING")) {
      const double tq1 = now_ms();
      GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
      fprintf(stderr, "[gh_ba] device pair lists: enqueue %.2f ms, then %.2f ms until the stream is idle (ub %zu pairs, %zu blocks)\n",
              tq1 - tq0, now_ms() - tq1, pairs_ub, blocks_ub);
    }
  }
  if (nblocks > 0 && (!device_pairs || pairs_check)) {
    int32_t *d_pa, *d_pb, *d_bs, *d_bci, *d_bcj, *d_sb, *d_sf;
    GH_TRY(db.upload(&d_sb, (const int32_t*)seg_blk.data(), seg_blk.size()));
    GH_TRY(db.upload(&d_sf, (const int32_t*)seg_first.data(), seg_first.size()));
    GH_TRY(db.upload(&d_pa, (const int32_t*)pair_a.data(), pair_a.size()));
    GH_TRY(db.upload(&d_pb, (const int32_t*)pair_b.data(), pair_b.size()));
    GH_TRY(db.upload(&d_bs, (const int32_t*)bstart.data(), bstart.size()));
    GH_TRY(db.upload(&d_bci, (const int32_t*)bci.data(), bci.size()));
    GH_TRY(db.upload(&d_bcj, (const int32_t*)bcj.data(), bcj.size()));
    SB_host = SchurBlocks{d_pa, d_pb, d_bs, d_bci, d_bcj, d_sb, d_sf, nblocks, nsegs};
    if (!device_pairs) SB = SB_host;
  }
  const double t_u3 = now_ms();
  if (getenv("GSLAM_HIP_BA_TIMING"))
    fprintf(stderr, "[gh_ba] upload phase: reserve %.2f, index lists up %.2f, chunk tables %.2f, pair lists %.2f ms\n", t_u0 - t_lists,
            t_u1 - t_u0, t_u2 - t_u1, t_u3 - t_u2);
  GH_TRY(db.alloc(&d_Hcc, (size_t)nc * 36));
  GH_TRY(db.alloc(&d_gc, (size_t)n));
  GH_TRY(db.alloc(&d_Hpp, (size_t)np * 9));
  GH_TRY(db.alloc(&d_gp, (size_t)np * 3));
  GH_TRY(db.alloc(&d_Hpi, (size_t)np * 9));
  GH_TRY(db.alloc(&d_dc, (size_t)n));
  GH_TRY(db.alloc(&d_dp, (size_t)np * 3));
  GH_TRY(db.alloc(&d_work, (size_t)n));
  GH_TRY(db.alloc(&d_dinv, (size_t)gh_div_up(n, 64) * 4096));
  GH_TRY(db.alloc(&d_W, (size_t)no * 18));
  GH_TRY(db.alloc(&d_xwork, (size_t)2 * 64 * (n + 1)));
  GH_TRY(db.alloc(&d_xh, (size_t)gh_div_up(n, 64) * 64));
  d_flow = nullptr;  // state of the single-launch factorisation (null when the shape does not fit it)
  if (const size_t words = gh_potrf_flow_words(ctx, n, 1)) GH_TRY(db.alloc(&d_flow, words));
  {
    // The reduced camera system couples two cameras only if they see a common point: with every point seen from cameras
    // at most `span` indices apart (a trajectory), S is a band of half-width 6 span + 5 and the band solver applies.
    const int span = S.cam_span;  // (found with the argument check)
    int want = ctx->ba_solver;
    if (const char* e = getenv("GSLAM_HIP_BA_SOLVER")) want = e[0] == 'd' ? 1 : (e[0] == 'b' ? 2 : 0);
    (void)want;
    cr_T = planned_tiles();
    if (n_bpts > 0 && cr_T == 0)  // (ba_order.hip only proposes a point border to the band solver: the dense assembly has camera columns only)
      return gh_set_error(ctx, GH_ERR_ARG, "gh_ba_solve: a point border without the band solver (internal)");
    d_cr_dinv = d_cr_W = S.d_arrow_ws = nullptr;
    S.map = CrMap{};
    if (cr_T) {
      // COMPACT columns: the band rows, one slot per reduction level for the fill, the border rows, the right-hand side --
      // 0.93 GB at C5 where the dense lower triangle takes 28.8 GB (GSLAM_HIP_BA_DENSE_S=1: the dense layout of rounds 4-5, A/B)
      const char* de = getenv("GSLAM_HIP_BA_DENSE_S");
      if (!(de && de[0] == '1')) {
        int brow = 0;
        lda = gh_cr_compact_lda(n_band, cr_T, n - n_band, &brow);
        S.map.m = 64 * cr_T;
        S.map.n_band = n_band;
        S.map.brow = brow;
        S.map.levels = brow / (64 * cr_T) - 2;
      }
    }
    S.lda = lda;
    GH_TRY(db.alloc(&d_S, (size_t)n * lda));
    if (cr_T) {
      GH_TRY(db.alloc(&d_cr_dinv, gh_cr_dinv_doubles(n_band, cr_T)));
      GH_TRY(db.alloc(&d_cr_W, gh_cr_panel_doubles(n_band, cr_T)));
      // (band-only systems too: the reduction's last levels go to the dense path -- chol_cr.hip, DENSE TOP)
      GH_TRY(db.alloc(&S.d_arrow_ws, gh_arrow_ws_doubles(ctx, n_band, cr_T, n - n_band)));
      S.d_border_nz = nullptr;
      if (n_band < n && !S.border_cam_nz.empty()) {
        // per (superblock, strip), through the levels of the reduction: the border kernels skip what stays zero
        const int m = 64 * cr_T, N = gh_div_up(n_band, m), nbs = gh_div_up(n - n_band, 16), ncb = nc - S.n_border;
        std::vector<uint8_t> init((size_t)N * nbs, 0), sym(gh_cr_border_symbolic_bytes(n_band, cr_T, n - n_band));
        for (int t = 0; t < nbs; ++t)
          for (int c = 0; c < ncb; ++c)
            if (S.border_cam_nz[(size_t)t * ncb + c]) {
              init[(size_t)((6 * c) / m) * nbs + t] = 1;
              init[(size_t)((6 * c + 5) / m) * nbs + t] = 1;
            }
        gh_cr_border_symbolic(n_band, cr_T, n - n_band, init.data(), sym.data());
        GH_TRY(db.alloc(&S.d_border_nz, sym.size()));
        GH_HIP(ctx, hipMemcpy(S.d_border_nz, sym.data(), sym.size(), hipMemcpyHostToDevice));
        if (opt.verbose) {
          size_t nzc = 0;
          for (size_t e = 0; e < (size_t)N * nbs; ++e) nzc += sym[e] != 0;
          fprintf(stderr, "[gh_ba] border structure: %zu of %zu (superblock, 16-row strip) blocks can be non-zero when eliminated\n", nzc, (size_t)N * nbs);
        }
      }
    }
    if (opt.verbose)
      fprintf(stderr, "[gh_ba] band cameras of a point at most %d indices apart, %d border cameras: half-bandwidth %d of n = %d -> %s\n",
              span, S.n_border, 6 * span + 5, n_band,
              cr_T ? (n_band < n ? "arrowhead solver (block cyclic reduction + dense border)" : "band solver (block cyclic reduction)")
                   : "dense factorisation");
  }
  GH_TRY(db.alloc(&d_cpart, (size_t)nchunks * 27));
  GH_TRY(db.alloc(&d_cpart2, (size_t)nchunks * 27));
  GH_TRY(db.alloc(&d_W2, (size_t)no * 18));
  GH_TRY(db.alloc(&d_Hpp2, (size_t)np * 9));
  GH_TRY(db.alloc(&d_gp2, (size_t)np * 3));
  GH_TRY(db.alloc(&S.d_keep, 2));
  d_spart = nullptr;
  if (!device_pairs) GH_TRY(db.alloc(&d_spart, (size_t)nsegs * 42));
  GH_TRY(db.alloc(&d_partial, (size_t)eval_blocks * 2));
  GH_TRY(db.alloc(&d_out, 4));
  GH_TRY(db.alloc(&d_gmax, 1));
  GH_TRY(db.alloc(&d_bad, 1));
  GH_TRY(db.alloc(&d_info, 1));
  S.nc = nc;
  S.np = np;
  S.no = no;
  S.deterministic = opt.deterministic;
  S.has_info = pr->obs_info != nullptr;
  S.has_pfree = pr->point_free != nullptr;
  } else if (pr) {  // ============================================================== resident graph: new values only
    auto up = [&](void* dst, const void* src, size_t bytes) -> gh_status {
      if (bytes) GH_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
      return GH_OK;
    };
    GH_TRY(up(d_poses, pr->cam_pose, (size_t)nc * 7 * 8));
    GH_TRY(up(d_pts, pr->point_xyz, (size_t)np * 3 * 8));
    GH_TRY(up(d_dof, pr->cam_dof, (size_t)nc * 4));
    if (pr->point_free) GH_TRY(up(d_pfree, pr->point_free, (size_t)np));
    GH_TRY(up(d_oxy, pr->obs_xy, (size_t)no * 16));
    if (pr->obs_info) GH_TRY(up(d_oinfo, pr->obs_info, (size_t)no * 32));
  }

  Problem P{nc, np, no, d_poses, d_dof, d_pts, d_pfree, d_ocam, d_opt, d_oxy, d_oinfo,
            d_pstart, d_plist, d_cstart, d_clist, opt.huber_delta};
  PointBorder PB;
  PB.n_pts = n_bpts;
  PB.pts = S.d_bpts;
  PB.slot = S.d_pbslot;
  PB.Hpp = d_Hpp;
  PB.gp = d_gp;

  auto eval_cost = [&](const double* poses_eval, const double* pts_eval, int with_model) -> gh_status {
    GH_LAUNCH(ctx, "ba_eval", eval_kernel, dim3(eval_blocks), dim3(256), 0, P, poses_eval, pts_eval, d_dc, d_dp,
              with_model, d_partial);
    GH_LAUNCH(ctx, "ba_reduce", reduce_final_kernel, dim3(1), dim3(256), 0, d_partial, eval_blocks, d_out);
    GH_HIP(ctx, hipMemcpyAsync(&rb->cost, d_out, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GH_OK;
  };

  GH_HIP(ctx, hipMemsetAsync(d_gmax, 0, sizeof(unsigned long long), ctx->stream));
  GH_HIP(ctx, hipMemsetAsync(d_bad, 0, sizeof(int), ctx->stream));
  double h2[2];
  const double t_upload = now_ms();
  GH_TRY(eval_cost(d_poses, d_pts, 0));
  h2[0] = rb->cost;
  h2[1] = rb->model;
  if (device_pairs && !S.ready) {  // the lists were built behind the uploads; their sizes came back with the first cost
    {
      if (pairs_check) {
        auto same = [&](const int32_t* dev, const std::vector<int32_t>& host, size_t count, const char* what) -> gh_status {
          std::vector<int32_t> got(count);
          GH_HIP(ctx, hipMemcpy(got.data(), dev, count * sizeof(int32_t), hipMemcpyDeviceToHost));
          for (size_t i = 0; i < count; ++i)
            if (got[i] != host[i])
              return gh_set_error(ctx, GH_ERR_NUMERIC, "device-built %s differs from the host list at %zu: %d vs %d", what, i,
                                  got[i], host[i]);
          return GH_OK;
        };
        if (rb->pair_counts[0] != (int)pair_a.size() || rb->pair_counts[1] != (int)bci.size() || rb->pair_counts[2] != nsegs)
          return gh_set_error(ctx, GH_ERR_NUMERIC, "device-built pair lists: %d pairs / %d blocks / %d segments, host %zu / %zu / %d",
                              rb->pair_counts[0], rb->pair_counts[1], rb->pair_counts[2], pair_a.size(), bci.size(), nsegs);
        GH_TRY(same(SB.pair_a, pair_a, pair_a.size(), "pair_a"));
        GH_TRY(same(SB.pair_b, pair_b, pair_b.size(), "pair_b"));
        GH_TRY(same(SB.bstart, bstart, bstart.size(), "bstart"));
        GH_TRY(same(SB.bci, bci, bci.size(), "bci"));
        GH_TRY(same(SB.bcj, bcj, bcj.size(), "bcj"));
        std::vector<int32_t> sf((size_t)nblocks + 1), sb((size_t)nsegs);
        GH_HIP(ctx, hipMemcpy(sf.data(), SB_host.seg_first, sf.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        GH_HIP(ctx, hipMemcpy(sb.data(), SB_host.seg_blk, sb.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        GH_TRY(same(SB.seg_first, sf, sf.size(), "seg_first"));
        GH_TRY(same(SB.seg_blk, sb, sb.size(), "seg_blk"));
      }
      nblocks = rb->pair_counts[1];
      nsegs = rb->pair_counts[2];
      SB.nblocks = nblocks;
      SB.nsegs = nsegs;
    }
    GH_TRY(db.alloc(&d_spart, (size_t)nsegs * 42));
    S.n_pairs = (size_t)rb->pair_counts[0];
  } else if (!S.ready) {
    S.n_pairs = pair_a.size();
  }
  if (opt.verbose)
    fprintf(stderr, "[gh_ba] %s: index lists %.2f ms (csr %.2f; %zu Schur pairs, %d blocks), upload %.2f ms, first cost %.2f ms\n",
            S.ready ? "resident graph" : "setup", t_lists - t_begin, t_csr - t_begin, S.n_pairs, nblocks, t_upload - t_lists,
            now_ms() - t_upload);
  S.ready = true;
  ctx->ba_last_solver = cr_T ? (n_band < n ? 3 : 2) : 1;
  ctx->ba_last_band_tiles = cr_T;
  ctx->ba_last_cam_span = S.cam_span;
  ctx->ba_last_border_cams = S.n_border;
  ctx->ba_last_border_points = n_bpts;
  ctx->ba_last_reordered = S.reordered;
  double cost = h2[0];
  sum->initial_cost = cost;
  double radius = opt.initial_radius, decrease = 2.0;
  bool need_lin = true;
  bool cr_flow_ok = true;  // band / arrowhead solver: the dense top may use the single-launch kernels (until one of their waits expires)
  int term = 0, it = 0;
  // Speculative linearisation: behind the candidate cost's read-back the stream goes on to linearise the CANDIDATE state
  // into the second buffer set while the host waits for that cost (an event, not the stream) and decides -- the ~30 us of
  // host turnaround per iteration are then GPU work the next iteration needs anyway if the step is accepted (most are; a
  // rejected step's speculation is dropped).  Same kernel, same inputs: the LM trace is bit-identical.
  const bool speculate = [] { const char* e = getenv("GSLAM_HIP_BA_SPECULATE"); return !(e && e[0] == '0'); }() && (np > 0 || nchunks > 0);
  bool spec_ready = false;  // the second buffer set holds the linearisation of what is now the current state
  unsigned stamp = 0;
  rb->stamp = 0;  // (nothing is in flight: the first cost above ended with a stream synchronisation)
  const double t_loop = now_ms();
  for (it = 0; it < opt.max_iterations; ++it) {
    if (need_lin) {  // (d_gmax and d_bad are zero here: cleared before the loop and by every backsub_update launch)
      if (spec_ready) {
        std::swap(d_W, d_W2);
        std::swap(d_Hpp, d_Hpp2);
        std::swap(d_gp, d_gp2);
        std::swap(d_cpart, d_cpart2);
      } else if (np > 0 || nchunks > 0) {
        GH_LAUNCH(ctx, "ba_lin", lin_kernel, dim3(nchunks + gh_div_up(np, 256)), dim3(256), 0, P, CC, d_cpart, d_W, d_Hpp, d_gp,
                  d_gmax);
      }
      // (the camera-side reduction shares its launch with the point-side damping below)
    }
    spec_ready = false;
    const bool fresh_lin = need_lin;
    need_lin = false;
    const bool publish = speculate && it + 1 < opt.max_iterations;  // this iteration's verdict goes to the host by reduce_publish_kernel
    if (fresh_lin) {
      const int nrb = gh_div_up(nc, 8);
      GH_LAUNCH(ctx, "ba_damp_points", lin_reduce_damp_kernel, dim3(nrb + gh_div_up(np, 256)), dim3(256), 0, nc, CC,
                (const double*)d_cpart, d_Hcc, d_gc, d_gmax, nrb, np, (const double*)d_Hpp, radius, d_Hpi, d_bad);
      // the gradient test is evaluated at the iteration's single synchronisation point below; if it fires, the step
      // computed meanwhile is simply dropped (same decisions as testing here, one host round trip less)
      if (!publish) GH_HIP(ctx, hipMemcpyAsync(&rb->gmax_bits, d_gmax, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    } else if (np > 0) {
      GH_LAUNCH(ctx, "ba_damp_points", damp_points_kernel, dim3(gh_div_up(np, 256)), dim3(256), 0, np, d_Hpp, radius,
                d_Hpi, d_bad);
    }
    PB.Hpp = d_Hpp;  // (the linearisation's buffer sets swap when a speculative linearisation is adopted)
    PB.gp = d_gp;
    const bool slim_init = d_flow != nullptr || cr_T != 0;  // both solvers read the lower tiles only
    const bool fused_seed = slim_init && no > 0 && opt.deterministic;  // then the seed rides with the segment sums below
    bool solve_state_ready = false;  // set when schur_reduce_kernel has cleared what the single-launch solve kernels need
    if (slim_init) {
      if (!fused_seed)
      {
        const int gx = cr_T ? 1 : gh_div_up(n + 1, 2048);
        GH_LAUNCH(ctx, "ba_schur_diag", schur_init_kernel, dim3((unsigned)gx * (unsigned)n), dim3(256), 0, n, lda, d_Hcc,
                  d_gc, radius, d_S, d_dc, 64 * cr_T, n_band, gx, S.map, PB);
      }
    } else {
      int pend = gh_prof_begin(ctx, "ba_schur_zero");
      hipError_t me = hipMemsetAsync(d_S, 0, (size_t)n * lda * sizeof(double), ctx->stream);
      gh_prof_end(ctx, pend);
      GH_HIP(ctx, me);
      GH_LAUNCH(ctx, "ba_schur_diag", schur_diag_kernel, dim3(nc), dim3(64), 0, nc, d_Hcc, d_gc, radius, d_S, lda, d_dc);
    }
    if (no > 0) {
      if (opt.deterministic) {
        const unsigned nsb = 8u * (unsigned)gh_div_up(nsegs, 32);
        if (fused_seed) {
          const int gx = cr_T ? 1 : gh_div_up(n + 1, 2048);  // (the band solver's seed: one workgroup per column)
          GH_LAUNCH(ctx, "ba_schur_blocks", schur_blocks_init_kernel, dim3(nsb + (unsigned)gx * (unsigned)n), dim3(256), 0, P,
                    SB, d_Hpi, d_gp, (const double*)d_W, d_spart, nsb, n, lda, (const double*)d_Hcc, (const double*)d_gc,
                    radius, d_S, d_dc, gx, 64 * cr_T, n_band, S.map, PB);
        } else {
          GH_LAUNCH(ctx, "ba_schur_blocks", schur_blocks_kernel, dim3(nsb), dim3(256), 0, P, SB, d_Hpi, d_gp,
                    (const double*)d_W, d_spart);
        }
        {
          const int nred = gh_div_up(nblocks, 4);
          SolveState st{nullptr, 0u, nullptr, 0u, d_info};
          if (d_flow) {
            st.flow_flags = d_flow;
            st.n_flow = (unsigned)gh_potrf_flow_flag_words(n, 1);
          }
          if (d_xh) {
            st.xh_words = reinterpret_cast<unsigned*>(d_xh);
            st.n_xh = (unsigned)gh_div_up(n, 64) * 64u * 2u;
          }
          const unsigned words = st.n_flow > st.n_xh ? st.n_flow : st.n_xh;
          GH_LAUNCH(ctx, "ba_schur_reduce", schur_reduce_kernel, dim3(nred + gh_div_up((int)(words ? words : 1u), 256)),
                    dim3(256), 0, SB, (const double*)d_spart, d_S, lda, d_dc, n, nred, st, S.map);
          solve_state_ready = true;
        }
      } else {
        GH_LAUNCH(ctx, "ba_schur_atomic", schur_atomic_kernel, dim3(gh_div_up(no, 256)), dim3(256), 0, P, d_Hpi, d_gp,
                  d_S, lda, d_dc, (const double*)d_W, S.map, (const int32_t*)S.d_pbslot);
      }
    }
    if (n_bpts > 0)  // E of the point border: W^T of the border points' observations into the border rows of their cameras' columns
      GH_LAUNCH(ctx, "ba_border_points", border_points_kernel, dim3(n_bpts), dim3(64), 0, P, PB, (const double*)d_W, d_S, lda, n_band,
                S.map);
    const double t_solve0 = now_ms();
    // (one single-launch factorisation at a time per process, until this iteration's synchronisation: see chol.hip)
    std::unique_lock<std::mutex> flow_lock(gh_potrf_flow_mutex(ctx->device), std::defer_lock);
    if (d_flow || cr_T) flow_lock.lock();  // (band / arrowhead: the dense top may run as the single-launch factorisation)
    // The whole candidate step is enqueued without waiting for the factorisation flags (the kernels have no
    // data-dependent control flow, so a failed factorisation only produces numbers that are then ignored): one host
    // synchronisation per iteration instead of three.
    // rhs -> row n of S: already there when schur_init_kernel + schur_reduce_kernel wrote it
    if (!(slim_init && opt.deterministic))
      GH_LAUNCH(ctx, "ba_rhs_row", rhs_to_row_kernel, dim3(gh_div_up(n, 256)), dim3(256), 0, d_dc, d_S, lda, n, S.map);
    if (cr_T) {  // (n_band == n: a band without a border)
      GH_TRY(gh_arrow_solve_dev_impl(ctx, d_S, n_band, n - n_band, lda, cr_T, d_cr_dinv, d_cr_W, S.d_arrow_ws, d_dc, d_info,
                                     solve_state_ready, cr_flow_ok, S.d_border_nz, S.map.m != 0));
    } else {
    GH_TRY(gh_potrf_dev_impl(ctx, d_S, n, lda, d_info, 1, d_dinv, d_xwork, d_flow, false, solve_state_ready));
    // y = L^-1 b is row n of the factored matrix; the back-substitution reads it in place
    GH_TRY(gh_potrs_bwd_dev_impl(ctx, d_S, n, lda, d_dc, d_work, d_dinv, d_S + n, lda, d_xh, d_info, solve_state_ready));
    }
    if (!publish) {
      GH_HIP(ctx, hipMemcpyAsync(&rb->info, d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      GH_HIP(ctx, hipMemcpyAsync(&rb->bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    }
    GH_LAUNCH(ctx, "ba_backsub", backsub_update_kernel, dim3(gh_div_up(nc > np ? nc : np, 256)), dim3(256), 0, P, d_Hpi,
              d_gp, d_dc, d_dp, (const double*)d_W, d_poses_new, d_pts_new, d_bad, d_gmax, publish ? S.d_keep : nullptr,
              (const int32_t*)S.d_pbslot, n_band);
    bool spec_launched = false;
    if (publish) {
      const unsigned want = ++stamp;
      GH_LAUNCH(ctx, "ba_eval", eval_kernel, dim3(eval_blocks), dim3(256), 0, P, (const double*)d_poses_new,
                (const double*)d_pts_new, d_dc, d_dp, 1, d_partial);
      GH_LAUNCH(ctx, "ba_reduce", reduce_publish_kernel, dim3(1), dim3(256), 0, (const double*)d_partial, eval_blocks,
                (const unsigned long long*)S.d_keep, (const int*)d_info, rb, want);
      Problem Pn = P;
      Pn.poses = d_poses_new;
      Pn.pts = d_pts_new;
      GH_LAUNCH(ctx, "ba_lin", lin_kernel, dim3(nchunks + gh_div_up(np, 256)), dim3(256), 0, Pn, CC, d_cpart2, d_W2, d_Hpp2,
                d_gp2, d_gmax);  // (d_gmax was cleared by backsub_update above: exactly what a linearisation at the top would find)
      spec_launched = true;
      // the iteration's one synchronisation: the verdict's stamp in pinned memory (bounded: a fault in a kernel must not hang us)
      {
        // Spin with `pause` for about 2 ms (a C4 iteration is 0.6 ms: the stamp is there within the spin and the host adds no
        // latency), then back off -- yield, and from 20 ms on sleep 50 us per poll: an iteration of a large dense problem (C5: 1 s)
        // no longer pins a host core at 100 % for its whole length.
        const double t_spin = now_ms();
        unsigned spins = 0;
        int phase = 0;  // 0 pause, 1 yield, 2 sleep
        while (rb->stamp != want) {
          if (phase == 0) {
            __builtin_ia32_pause();
            if ((++spins & 0x3FFu) == 0u && now_ms() - t_spin > 2.0) phase = 1;
          } else {
            const double waited = now_ms() - t_spin;
            if (waited > 10000.0) break;
            if (phase == 1) {
              std::this_thread::yield();
              if (waited > 20.0) phase = 2;
            } else {
              std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
          }
        }
        if (rb->stamp != want) {
          GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
          if (rb->stamp != want) return gh_set_error(ctx, GH_ERR_HIP, "gh_ba_solve: the iteration's verdict never reached the host");
        }
        std::atomic_thread_fence(std::memory_order_acquire);
      }
    } else {
      GH_TRY(eval_cost(d_poses_new, d_pts_new, 1));  // the iteration's one synchronisation
    }
    if (flow_lock.owns_lock()) flow_lock.unlock();
    h2[0] = rb->cost;
    h2[1] = rb->model;
    sum->solve_ms_total += now_ms() - t_solve0;
    if (fresh_lin) {
      double gmax;
      memcpy(&gmax, &rb->gmax_bits, sizeof(double));
      if (gmax <= opt.gradient_tolerance) {
        term = 2;
        break;
      }
    }
    if (rb->info > n && (cr_T ? cr_flow_ok : (d_flow || d_xh))) {
      // A bounded wait inside one of the single-launch kernels expired (chol.hip: the launch could not get all its
      // workgroups resident, e.g. another process holds part of the GPU).  Nothing was decided yet: repeat this iteration
      // on the launch-per-step path and stay on it for the rest of the solve.
      if (opt.verbose) fprintf(stderr, "[gh_ba] it %3d: single-launch solve timed out (info %d), repeating on the launch path\n", it, rb->info);
      d_flow = nullptr;
      d_xh = nullptr;
      cr_flow_ok = false;  // (band / arrowhead: the dense top stays off the single-launch kernels)
      if (fresh_lin) need_lin = true;  // (cheap, and keeps the gradient read-back of this iteration in place)
      if (spec_launched) {  // the speculation wrote the gradient maximum the repeated linearisation accumulates into
        GH_HIP(ctx, hipMemsetAsync(d_gmax, 0, sizeof(unsigned long long), ctx->stream));
      }
      --it;
      continue;
    }
    const bool ok = rb->info == 0 && rb->bad == 0;
    double new_cost = cost, model = 0, rho = -1;
    if (ok) {
      new_cost = h2[0];
      model = h2[1];
      rho = model > 0 ? (cost - new_cost) / model : -1;
      if (!(new_cost == new_cost)) rho = -1;  // NaN guard
    }
    const bool acc = ok && rho > opt.min_relative_decrease;
    if (sum->trace_len < GH_BA_MAX_TRACE) {
      sum->trace_cost[sum->trace_len] = new_cost;
      sum->trace_radius[sum->trace_len] = radius;
      sum->trace_accepted[sum->trace_len] = (uint8_t)acc;
      sum->trace_len++;
    }
    if (opt.verbose)
      fprintf(stderr, "[gh_ba] it %3d cost %.9e -> %.9e model %.3e rho %.3f radius %.3e %s\n", it, cost, new_cost,
              model, rho, radius, acc ? "accepted" : (ok ? "rejected" : "solve failed"));
    if (acc) {
      const double dcost = cost - new_cost;
      std::swap(d_poses, d_poses_new);
      std::swap(d_pts, d_pts_new);
      P.poses = d_poses;
      P.pts = d_pts;
      const double t = 2.0 * rho - 1.0;
      radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
      if (radius > 1e16) radius = 1e16;
      decrease = 2.0;
      sum->accepted++;
      need_lin = true;
      spec_ready = spec_launched;
      const double prev = cost;
      cost = new_cost;
      if (fabs(dcost) <= opt.function_tolerance * prev) {
        term = 1;
        ++it;
        break;
      }
    } else {
      radius = radius / decrease;
      decrease *= 2.0;
      if (radius < 1e-32) {
        term = 3;
        ++it;
        break;
      }
    }
  }
  sum->iterations = it;
  sum->termination = term;
  sum->final_cost = cost;
  const double t_loop_end = now_ms();
  if (download && pr) {
    GH_HIP(ctx, hipMemcpyAsync(pr->cam_pose, d_poses, (size_t)nc * 7 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (np > 0)
      GH_HIP(ctx, hipMemcpyAsync(pr->point_xyz, d_pts, (size_t)np * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  }
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  sum->total_ms = now_ms() - t_begin;
  if (getenv("GSLAM_HIP_BA_TIMING"))
    fprintf(stderr, "[gh_ba] whole solve %.2f ms: lists + early upload %.2f, to the first cost %.2f, first cost %.2f, iterations %.2f, result download %.2f\n",
            sum->total_ms, t_lists - t_begin, t_upload - t_lists, t_loop - t_upload, t_loop_end - t_loop, now_ms() - t_loop_end);
  return term == 3 ? GH_ERR_NUMERIC : GH_OK;
}

// (the camera order -- band / arrow / bandwidth-reducing -- lives in ba_order.hip)
// The caller's problem in arrow order: cam_pose / cam_dof / obs_cam are re-indexed copies, everything else is shared.
struct ArrowProblem {
  gh_ba_problem pr;
  std::vector<double>& pose;     // (the context's buffers: see common.h)
  std::vector<int32_t>&dof, &ocam;
  explicit ArrowProblem(gh_ctx* ctx) : pose(ctx->ba_order_pose), dof(ctx->ba_order_dof), ocam(ctx->ba_order_ocam) {}
  void build(const gh_ba_problem* src, const std::vector<int32_t>& perm) {
    pr = *src;
    const int nc = src->n_cams, no = src->n_obs;
    pose.resize((size_t)nc * 7);
    dof.resize((size_t)nc);
    std::vector<int32_t> inv((size_t)nc);
    for (int c = 0; c < nc; ++c) {
      inv[perm[c]] = c;
      memcpy(&pose[(size_t)c * 7], src->cam_pose + (size_t)perm[c] * 7, 7 * sizeof(double));
      dof[c] = src->cam_dof[perm[c]];
    }
    ocam.resize((size_t)no);
    HostPool& pool = HostPool::get();
    const int T = no >= (1 << 20) ? std::min(pool.size(), 8) : 1;
    pool.run(T, [&](int t) {
      const int k0 = (int)((long long)no * t / T), k1 = (int)((long long)no * (t + 1) / T);
      for (int k = k0; k < k1; ++k) ocam[k] = inv[src->obs_cam[k]];
    });
    pr.cam_pose = pose.data();
    pr.cam_dof = dof.data();
    pr.obs_cam = ocam.data();
  }
};
bool ba_arrow_wanted(const gh_ctx* ctx) {
  int want = ctx->ba_solver;
  if (const char* e = getenv("GSLAM_HIP_BA_SOLVER")) want = e[0] == 'd' ? 1 : (e[0] == 'b' ? 2 : 0);
  if (const char* e = getenv("GSLAM_HIP_BA_ARROW")) if (e[0] == '0') return false;  // A/B measurements: loop closures -> dense, as in rounds 1-4
  return want != 1;
}
// the camera order of a new graph (ba_order.hip): S.perm / S.n_border / S.reordered
void ba_choose_order(const gh_ctx* ctx, BaSession& S, const gh_ba_problem* pr) {
  if (!(pr->n_cams > 0 && pr->cam_pose && pr->cam_dof && pr->obs_cam && pr->obs_point && ba_arrow_wanted(ctx))) return;
  const char* e = getenv("GSLAM_HIP_BA_REORDER");  // "0": the caller's camera order as it is (rounds 4-5; A/B measurements)
  const double t0 = now_ms();
  S.n_border = gh_ba_order_cameras(pr, S.perm, &S.reordered, !(e && e[0] == '0'), &S.span_measured, &S.border_pts);
  if (getenv("GSLAM_HIP_BA_TIMING"))
    fprintf(stderr, "[gh_ba] camera order: %.2f ms (%s, %d border cameras)\n", now_ms() - t0,
            S.reordered ? "bandwidth-reducing order applied" : (S.perm.empty() ? "caller's order" : "arrow order"), S.n_border);
  if (getenv("GSLAM_HIP_BA_TIMING") && !S.border_pts.empty())
    fprintf(stderr, "[gh_ba] border: %zu long-range points stay out of the Schur complement\n", S.border_pts.size());
}

}  // namespace

extern "C" gh_status gh_ba_solve(gh_ctx* ctx, gh_ba_problem* pr, const gh_ba_options* opt_in, gh_ba_summary* sum_out) {
  if (!ctx || !pr) return GH_ERR_ARG;
  GH_ENTER(ctx);
  BaSession S;
  S.db = new (std::nothrow) DevBuf(ctx);
  if (!S.db) return GH_ERR_NOMEM;
  // (summary.total_ms covers the whole call: the solver's own camera ordering and the renumbered copy of the problem included)
  const double t_call = now_ms();
  gh_ba_summary local_sum;
  if (!sum_out) sum_out = &local_sum;
  ba_choose_order(ctx, S, pr);
  if (S.perm.empty()) {
    const double t_order = now_ms() - t_call;
    const gh_status st = ba_run(ctx, S, pr, opt_in, sum_out, true);
    sum_out->total_ms += t_order;
    return st;
  }
  ArrowProblem ap(ctx);
  ap.build(pr, S.perm);
  const double t_order = now_ms() - t_call;
  const gh_status st = ba_run(ctx, S, &ap.pr, opt_in, sum_out, true);
  const double t_back = now_ms();
  for (int c = 0; c < pr->n_cams; ++c) memcpy(pr->cam_pose + (size_t)S.perm[c] * 7, &ap.pose[(size_t)c * 7], 7 * sizeof(double));
  sum_out->total_ms += t_order + (now_ms() - t_back);
  if (getenv("GSLAM_HIP_BA_TIMING")) fprintf(stderr, "[gh_ba] camera order + renumbered problem: %.2f ms of the call\n", t_order);
  return st;
}

// ---------------------------------------------------------------- device-resident graph across solves
struct gh_ba_graph {
  gh_ctx* ctx = nullptr;
  BaSession S;
};

extern "C" gh_status gh_ba_graph_create(gh_ctx* ctx, const gh_ba_problem* problem, const gh_ba_options* options,
                                        gh_ba_graph** out) {
  if (!ctx || !problem || !out) return GH_ERR_ARG;
  GH_ENTER(ctx);
  *out = nullptr;
  gh_ba_graph* g = new (std::nothrow) gh_ba_graph();
  if (!g) return GH_ERR_NOMEM;
  g->ctx = ctx;
  g->S.db = new (std::nothrow) DevBuf(ctx, &g->S.arena, &g->S.arena_bytes);
  if (!g->S.db) {
    delete g;
    return GH_ERR_NOMEM;
  }
  gh_ba_options o;
  gh_ba_default_options(&o);
  if (options) o = *options;
  o.max_iterations = 0;  // set-up only: lists, tables, uploads, the initial cost
  gh_ba_summary sum;
  gh_ba_problem pr = *problem;
  ArrowProblem ap(ctx);
  ba_choose_order(ctx, g->S, &pr);
  if (!g->S.perm.empty()) {
    ap.build(problem, g->S.perm);
    pr = ap.pr;
  }
  const gh_status st = ba_run(ctx, g->S, &pr, &o, &sum, false);
  if (st != GH_OK) {
    delete g;
    return st;
  }
  *out = g;
  return GH_OK;
}

extern "C" void gh_ba_graph_destroy(gh_ba_graph* g) {
  if (!g) return;
  GH_ENTER(g->ctx);
  hipStreamSynchronize(g->ctx->stream);
  delete g;
}

extern "C" gh_status gh_ba_graph_update(gh_ba_graph* g, const double* cam_pose, const double* point_xyz, const double* obs_xy,
                                        const double* obs_info, const int32_t* cam_dof, const uint8_t* point_free) {
  if (!g) return GH_ERR_ARG;
  gh_ctx* ctx = g->ctx;
  GH_ENTER(ctx);
  BaSession& S = g->S;
  GH_CHECK_ARG(ctx, S.ready && (obs_info == nullptr || S.has_info) && (point_free == nullptr || S.has_pfree));
  auto up = [&](void* dst, const void* src, size_t bytes) -> gh_status {
    if (src && bytes) GH_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return GH_OK;
  };
  std::vector<double> pose_p;
  std::vector<int32_t> dof_p;
  if (!S.perm.empty()) {  // arrow order on the device: translate the caller's per-camera arrays
    if (cam_pose) {
      pose_p.resize((size_t)S.nc * 7);
      for (int c = 0; c < S.nc; ++c) memcpy(&pose_p[(size_t)c * 7], cam_pose + (size_t)S.perm[c] * 7, 56);
      cam_pose = pose_p.data();
    }
    if (cam_dof) {
      dof_p.resize((size_t)S.nc);
      for (int c = 0; c < S.nc; ++c) dof_p[c] = cam_dof[S.perm[c]];
      cam_dof = dof_p.data();
    }
  }
  // (every copy is enqueued whatever the one before returned, and the stream is drained before pose_p / dof_p go out of scope: a
  //  failing later copy must not leave an earlier asynchronous one reading freed memory)
  gh_status st = GH_OK;
  auto keep = [&](gh_status s) { if (st == GH_OK) st = s; };
  keep(up(S.d_poses, cam_pose, (size_t)S.nc * 56));
  keep(up(S.d_pts, point_xyz, (size_t)S.np * 24));
  keep(up(S.d_oxy, obs_xy, (size_t)S.no * 16));
  keep(up(S.d_oinfo, obs_info, (size_t)S.no * 32));
  keep(up(S.d_dof, cam_dof, (size_t)S.nc * 4));
  keep(up(S.d_pfree, point_free, (size_t)S.np));
  const hipError_t se = hipStreamSynchronize(ctx->stream);  // the caller's arrays may be pageable and freed on return
  if (st != GH_OK) return st;
  GH_HIP(ctx, se);
  return GH_OK;
}

extern "C" gh_status gh_ba_graph_solve(gh_ba_graph* g, const gh_ba_options* options, gh_ba_summary* summary) {
  if (!g) return GH_ERR_ARG;
  GH_ENTER(g->ctx);
  return ba_run(g->ctx, g->S, nullptr, options, summary, false);
}

extern "C" gh_status gh_ba_graph_read(gh_ba_graph* g, double* cam_pose, double* point_xyz) {
  if (!g) return GH_ERR_ARG;
  gh_ctx* ctx = g->ctx;
  GH_ENTER(ctx);
  std::vector<double> pose_p;
  double* pose_dst = cam_pose;
  if (cam_pose && !g->S.perm.empty()) {
    pose_p.resize((size_t)g->S.nc * 7);
    pose_dst = pose_p.data();
  }
  if (cam_pose)
    GH_HIP(ctx, hipMemcpyAsync(pose_dst, g->S.d_poses, (size_t)g->S.nc * 56, hipMemcpyDeviceToHost, ctx->stream));
  if (point_xyz && g->S.np > 0)
    GH_HIP(ctx, hipMemcpyAsync(point_xyz, g->S.d_pts, (size_t)g->S.np * 24, hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (cam_pose && !g->S.perm.empty())
    for (int c = 0; c < g->S.nc; ++c) memcpy(cam_pose + (size_t)g->S.perm[c] * 7, &pose_p[(size_t)c * 7], 56);
  return GH_OK;
}

// ---------------------------------------------------------------- pose-only LM in ONE launch
// optimizePnP / optimizePose are called once per frame by a tracking front end (GSLAM/core/Optimizer.h:193-207): one
// camera, a few hundred fixed points, ~10 LM iterations.  Through gh_ba_solve that is ~10 x (6 launches + a host round
// trip) plus the list building of a general graph: 0.5 ms of pure latency.  Here the whole LM loop runs inside one
// workgroup -- linearisation, the 6 x 6 damped solve, candidate cost, accept / reject, trust-region update -- with
// fixed-shape block reductions, and the host sees one upload, one launch and one download.  Same algorithm, same
// policy constants and the same per-observation arithmetic (linearize<>, se3_retract) as the general path and
// oracle/ba_oracle.c; only the order of the sums over the observations differs (within the parity tolerance).
namespace {

struct PnpResult {
  gh_ba_summary sum;
  double pose[7];
  double info[36];
};

// v[0..N) summed over the 256 threads of the block in a fixed order; the totals land in tot[0..N) (valid after return)
template <int N>
__device__ inline void pnp_block_sum(double* v, double (*red)[36], double* tot) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double x = v[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
    if (lane == 0) red[wv][i] = x;
  }
  __syncthreads();
  if (threadIdx.x < N) tot[threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
  __syncthreads();
}

__global__ __launch_bounds__(256) void pnp_lm_kernel(const double* __restrict__ X, const double* __restrict__ M, int n,
                                                     const double* __restrict__ pose_in, int dof, gh_ba_options opt,
                                                     int want_info, PnpResult* __restrict__ out) {
  __shared__ double s_pose[7], s_dc[6], s_red[4][36], s_tot[36];
  __shared__ double s_cost, s_radius, s_decrease;
  __shared__ int s_ctl[4];  // [0] stop, [1] solve ok, [2] need_lin
  const int tid = threadIdx.x;
  const double huber = opt.huber_delta;
  if (tid < 7) s_pose[tid] = pose_in[tid];
  __syncthreads();
  auto rho_of = [&](double s) { return (huber > 0 && s > huber * huber) ? 2.0 * huber * sqrt(s) - huber * huber : s; };
  // robust cost at `pose` (sum of rho, not yet halved); with `old`: infinite if an observation valid at `old` is lost
  auto cost_pass = [&](const double* pose, const double* old) {
    double c = 0;
    for (int k = tid; k < n; k += 256) {
      Obs o;
      const bool in_front = linearize<false>(pose, 0, X + 3 * k, 0, M + 2 * k, nullptr, huber, o);
      double ck = in_front ? rho_of(o.s) : 0.0;
      if (old != nullptr && !in_front && linearize<false>(old, 0, X + 3 * k, 0, M + 2 * k, nullptr, huber, o)) ck = __builtin_inf();
      c += ck;
    }
    return c;
  };
  double v[36];
  v[0] = cost_pass(s_pose, nullptr);
  pnp_block_sum<1>(v, s_red, s_tot);
  if (tid == 0) {
    s_cost = 0.5 * s_tot[0];
    s_radius = opt.initial_radius;
    s_decrease = 2.0;
    s_ctl[0] = 0;
    s_ctl[2] = 1;
    out->sum.initial_cost = s_cost;
    out->sum.trace_len = 0;
    out->sum.accepted = 0;
  }
  __syncthreads();
  double H[21], g[6];  // lower triangle of J^T L J (row a >= column b at a (a + 1) / 2 + b) and J^T L r: thread 0 keeps them
  int it = 0, term = 0;
  for (it = 0; it < opt.max_iterations; ++it) {
    if (s_ctl[2]) {
      for (int i = 0; i < 27; ++i) v[i] = 0;
      for (int k = tid; k < n; k += 256) {
        Obs o;
        if (!linearize<true>(s_pose, dof, X + 3 * k, 0, M + 2 * k, nullptr, huber, o)) continue;
        double L[4];
        weighted_info(nullptr, o.w, L);
        const double Lr[2] = {L[0] * o.r[0] + L[1] * o.r[1], L[2] * o.r[0] + L[3] * o.r[1]};
        double LJ[12];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          LJ[j] = L[0] * o.Jc[j] + L[1] * o.Jc[6 + j];
          LJ[6 + j] = L[2] * o.Jc[j] + L[3] * o.Jc[6 + j];
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          v[21 + a] += o.Jc[a] * Lr[0] + o.Jc[6 + a] * Lr[1];
#pragma unroll
          for (int b = 0; b <= a; ++b) v[a * (a + 1) / 2 + b] += o.Jc[a] * LJ[b] + o.Jc[6 + a] * LJ[6 + b];
        }
      }
      pnp_block_sum<27>(v, s_red, s_tot);
      if (tid == 0) {
        double gmax = 0;
        for (int i = 0; i < 21; ++i) H[i] = s_tot[i];
        for (int a = 0; a < 6; ++a) {
          g[a] = s_tot[21 + a];
          gmax = fmax(gmax, fabs(g[a]));
        }
        if (gmax <= opt.gradient_tolerance) s_ctl[0] = 2;  // termination 2, nothing recorded for this iteration
        s_ctl[2] = 0;
      }
      __syncthreads();
      if (s_ctl[0] == 2) {
        term = 2;
        break;
      }
    }
    if (tid == 0) {  // damped 6 x 6 system, Cholesky, dc = -(H + D)^-1 g
      const double radius = s_radius;
      double Lc[21];
      bool ok = true;
      for (int j = 0; j < 6 && ok; ++j) {
        const double hjj = H[j * (j + 1) / 2 + j];
        double d = hjj + (hjj < 1e-6 ? 1e-6 : (hjj > 1e32 ? 1e32 : hjj)) / radius;
        for (int k = 0; k < j; ++k) d -= Lc[j * (j + 1) / 2 + k] * Lc[j * (j + 1) / 2 + k];
        if (!(d > 0.0)) {
          ok = false;
          break;
        }
        const double ljj = sqrt(d);
        Lc[j * (j + 1) / 2 + j] = ljj;
        for (int i = j + 1; i < 6; ++i) {
          double x = H[i * (i + 1) / 2 + j];
          for (int k = 0; k < j; ++k) x -= Lc[i * (i + 1) / 2 + k] * Lc[j * (j + 1) / 2 + k];
          Lc[i * (i + 1) / 2 + j] = x / ljj;
        }
      }
      if (ok) {
        double y[6];
        for (int i = 0; i < 6; ++i) {
          double x = -g[i];
          for (int k = 0; k < i; ++k) x -= Lc[i * (i + 1) / 2 + k] * y[k];
          y[i] = x / Lc[i * (i + 1) / 2 + i];
        }
        for (int i = 5; i >= 0; --i) {
          double x = y[i];
          for (int k = i + 1; k < 6; ++k) x -= Lc[k * (k + 1) / 2 + i] * y[k];
          y[i] = x / Lc[i * (i + 1) / 2 + i];
        }
        for (int i = 0; i < 6; ++i) s_dc[i] = y[i];
      }
      s_ctl[1] = ok ? 1 : 0;
    }
    __syncthreads();
    const bool ok = s_ctl[1] != 0;
    double pnew[7];
    if (ok) {
      double dc[6];
      for (int a = 0; a < 6; ++a) dc[a] = s_dc[a];
      if ((dof & 63) == 0) {
        for (int i = 0; i < 7; ++i) pnew[i] = s_pose[i];
      } else {
        double cur[7];
        for (int i = 0; i < 7; ++i) cur[i] = s_pose[i];
        se3_retract(cur, dc, pnew);
      }
      double model = 0;
      for (int k = tid; k < n; k += 256) {
        Obs o;
        if (!linearize<true>(s_pose, dof, X + 3 * k, 0, M + 2 * k, nullptr, huber, o)) continue;
        double L[4];
        weighted_info(nullptr, o.w, L);
        double Jd[2] = {0, 0};
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          Jd[0] += o.Jc[a] * dc[a];
          Jd[1] += o.Jc[6 + a] * dc[a];
        }
        const double LJd[2] = {L[0] * Jd[0] + L[1] * Jd[1], L[2] * Jd[0] + L[3] * Jd[1]};
        const double Lr[2] = {L[0] * o.r[0] + L[1] * o.r[1], L[2] * o.r[0] + L[3] * o.r[1]};
        model -= Jd[0] * Lr[0] + Jd[1] * Lr[1] + 0.5 * (Jd[0] * LJd[0] + Jd[1] * LJd[1]);
      }
      v[0] = cost_pass(pnew, s_pose);
      v[1] = model;
      pnp_block_sum<2>(v, s_red, s_tot);
    }
    if (tid == 0) {
      const double cost = s_cost, radius = s_radius;
      double new_cost = cost, model = 0, rho = -1;
      if (ok) {
        new_cost = 0.5 * s_tot[0];
        model = s_tot[1];
        rho = model > 0 ? (cost - new_cost) / model : -1;
        if (!(new_cost == new_cost)) rho = -1;
      }
      const bool acc = ok && rho > opt.min_relative_decrease;
      const int tl = out->sum.trace_len;
      if (tl < GH_BA_MAX_TRACE) {
        out->sum.trace_cost[tl] = new_cost;
        out->sum.trace_radius[tl] = radius;
        out->sum.trace_accepted[tl] = (uint8_t)acc;
        out->sum.trace_len = tl + 1;
      }
      if (acc) {
        const double dcost = cost - new_cost;
        for (int i = 0; i < 7; ++i) s_pose[i] = pnew[i];
        const double t = 2.0 * rho - 1.0;
        double r2 = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
        if (r2 > 1e16) r2 = 1e16;
        s_radius = r2;
        s_decrease = 2.0;
        out->sum.accepted = out->sum.accepted + 1;
        s_ctl[2] = 1;
        s_cost = new_cost;
        if (fabs(dcost) <= opt.function_tolerance * cost) s_ctl[0] = 1;
      } else {
        s_radius = radius / s_decrease;
        s_decrease = s_decrease * 2.0;
        if (s_radius < 1e-32) s_ctl[0] = 3;
      }
    }
    __syncthreads();
    if (s_ctl[0] != 0) {
      term = s_ctl[0];
      ++it;
      break;
    }
  }
  if (want_info) {  // J^T W J at the solution (columns masked by dof), W = the Huber IRLS weight
    for (int i = 0; i < 36; ++i) v[i] = 0;
    for (int k = tid; k < n; k += 256) {
      Obs o;
      if (!linearize<true>(s_pose, dof, X + 3 * k, 0, M + 2 * k, nullptr, huber, o)) continue;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) v[6 * a + b] += o.w * (o.Jc[a] * o.Jc[b] + o.Jc[6 + a] * o.Jc[6 + b]);
    }
    pnp_block_sum<36>(v, s_red, s_tot);
    if (tid < 36) out->info[tid] = s_tot[tid];
  }
  if (tid == 0) {
    out->sum.iterations = it;
    out->sum.termination = term;
    out->sum.final_cost = s_cost;
    out->sum.solve_ms_total = 0;
    out->sum.total_ms = 0;
    for (int i = 0; i < 7; ++i) out->pose[i] = s_pose[i];
  }
}

}  // namespace

// Pose-only optimisation = the same solver on a 1-camera graph with every point fixed.
extern "C" gh_status gh_ba_pnp(gh_ctx* ctx, const double* points_xyz, const double* obs_xy, int n, double* pose,
                               int dof, const gh_ba_options* options, double* information_out,
                               gh_ba_summary* summary) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, n >= 0 && pose && (n == 0 || (points_xyz && obs_xy)));
  {
    gh_ba_options o;
    gh_ba_default_options(&o);
    if (options) o = *options;
    const char* pe = getenv("GSLAM_HIP_PNP_KERNEL");  // "0": through the general solver (A/B measurements, tests of both)
    const bool kernel_env = !(pe && pe[0] == '0');
    if (kernel_env && !o.verbose && n <= 65536) {
      // one pinned block up ([points | observations | pose]), one launch, one pinned block down
      const double t0 = now_ms();
      const size_t in_doubles = (size_t)5 * n + 7, in_bytes = (in_doubles * 8 + 255) & ~(size_t)255;
      char *hp = nullptr, *dp = nullptr;
      GH_TRY(gh_pinned(ctx, in_bytes + sizeof(PnpResult), (void**)&hp));
      GH_TRY(gh_scratch(ctx, in_bytes + sizeof(PnpResult), (void**)&dp));
      double* hin = reinterpret_cast<double*>(hp);
      if (n) {
        memcpy(hin, points_xyz, (size_t)3 * n * 8);
        memcpy(hin + (size_t)3 * n, obs_xy, (size_t)2 * n * 8);
      }
      memcpy(hin + (size_t)5 * n, pose, 56);
      GH_HIP(ctx, hipMemcpyAsync(dp, hp, in_doubles * 8, hipMemcpyHostToDevice, ctx->stream));
      const double* din = reinterpret_cast<const double*>(dp);
      PnpResult* dres = reinterpret_cast<PnpResult*>(dp + in_bytes);
      GH_LAUNCH(ctx, "ba_pnp_lm", pnp_lm_kernel, dim3(1), dim3(256), 0, din, din + (size_t)3 * n, n, din + (size_t)5 * n, dof, o,
                information_out ? 1 : 0, dres);
      PnpResult* hres = reinterpret_cast<PnpResult*>(hp + in_bytes);
      GH_HIP(ctx, hipMemcpyAsync(hres, dres, sizeof(PnpResult), hipMemcpyDeviceToHost, ctx->stream));
      GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
      memcpy(pose, hres->pose, 56);
      if (information_out) memcpy(information_out, hres->info, 36 * 8);
      if (summary) {
        *summary = hres->sum;
        summary->total_ms = now_ms() - t0;
      }
      return hres->sum.termination == 3 ? GH_ERR_NUMERIC : GH_OK;
    }
  }
  std::vector<int32_t> ocam((size_t)(n > 0 ? n : 1), 0), opt((size_t)(n > 0 ? n : 1));
  std::vector<uint8_t> pfree((size_t)(n > 0 ? n : 1), 0);
  std::vector<double> pts(points_xyz, points_xyz + (size_t)3 * n);
  for (int i = 0; i < n; ++i) opt[i] = i;
  int32_t d = dof;
  gh_ba_problem pr;
  memset(&pr, 0, sizeof(pr));
  pr.n_cams = 1;
  pr.n_points = n;
  pr.n_obs = n;
  pr.cam_pose = pose;
  pr.cam_dof = &d;
  pr.point_xyz = pts.data();
  pr.point_free = pfree.data();
  pr.obs_cam = ocam.data();
  pr.obs_point = opt.data();
  pr.obs_xy = obs_xy;
  gh_status st = gh_ba_solve(ctx, &pr, options, summary);
  if (st != GH_OK && st != GH_ERR_NUMERIC) return st;
  if (information_out) {
    // J^T J at the solution (host, n is small): same residual model as the kernels
    gh_ba_options o;
    gh_ba_default_options(&o);
    if (options) o = *options;
    for (int i = 0; i < 36; ++i) information_out[i] = 0;
    for (int k = 0; k < n; ++k) {
      const double* q = pose;
      const double* X = points_xyz + 3 * k;
      const double dd[3] = {X[0] - q[4], X[1] - q[5], X[2] - q[6]};
      const double qc[4] = {-q[0], -q[1], -q[2], q[3]};
      double uvx = qc[1] * dd[2] - qc[2] * dd[1], uvy = qc[2] * dd[0] - qc[0] * dd[2], uvz = qc[0] * dd[1] - qc[1] * dd[0];
      uvx += uvx; uvy += uvy; uvz += uvz;
      const double Xc[3] = {dd[0] + qc[3] * uvx + (qc[1] * uvz - qc[2] * uvy),
                            dd[1] + qc[3] * uvy + (qc[2] * uvx - qc[0] * uvz),
                            dd[2] + qc[3] * uvz + (qc[0] * uvy - qc[1] * uvx)};
      if (!(Xc[2] > 1e-9)) continue;
      const double iz = 1.0 / Xc[2], u = Xc[0] * iz, v = Xc[1] * iz;
      const double r0 = u - obs_xy[2 * k], r1 = v - obs_xy[2 * k + 1];
      const double s = r0 * r0 + r1 * r1;
      const double w = (o.huber_delta > 0 && s > o.huber_delta * o.huber_delta) ? o.huber_delta / sqrt(s) : 1.0;
      const double Pm[6] = {iz, 0, -u * iz, 0, iz, -v * iz};
      const double D[18] = {-1, 0, 0, 0, -Xc[2], Xc[1], 0, -1, 0, Xc[2], 0, -Xc[0], 0, 0, -1, -Xc[1], Xc[0], 0};
      double J[12];
      for (int a = 0; a < 2; ++a)
        for (int c = 0; c < 6; ++c) {
          double acc = 0;
          for (int j = 0; j < 3; ++j) acc += Pm[a * 3 + j] * D[j * 6 + c];
          J[a * 6 + c] = ((dof >> c) & 1) ? acc : 0.0;
        }
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) information_out[6 * a + b] += w * (J[a] * J[b] + J[6 + a] * J[6 + b]);
    }
  }
  return st;
}


// ---------------------------------------------------------------- Optimizer::magin: bundle graph -> pose graph
// GSLAM/core/Optimizer.h:230-232 ("Convert bundle graph to pose graph") has no implementation in the reference; the
// specification is the header of oracle_ba_marginalize (oracle/ba_oracle.c): for every pair of cameras (i < j) that share
// at least `min_shared` points, one SE3 edge whose information is that of camera j's pose RELATIVE to camera i held fixed,
// from the two-view problem over the shared points at the current estimate,
//     Lambda_ij = sum_p  A_jp - B_jp V_p^-1 B_jp^T,   A = Jc^T L Jc,  B = Jc^T L Jp  (observation of p in j),
//                                                     V_p = Jp^T L Jp of BOTH observations of p  (fixed point: A alone)
// in the perturbation T_j <- T_j exp(delta) the pose-graph residual log(M^-1 T_i^-1 T_j) is linear in.  The host lists the
// (first, second) observation pairs per camera pair (sorted, stable); one wave per camera pair sums its terms with the
// reduce-scatter butterfly of the Schur product: fixed order, deterministic.
namespace {
template <int K, int M>
__device__ __forceinline__ void scatter_step36(double (&v)[36], int lane, int& elem, int& mylen) {
  constexpr int H = (K + 1) / 2;
  const bool up = (lane & M) != 0;
#pragma unroll
  for (int j = 0; j < H; ++j) {
    const double lo = v[j], hi = H + j < K ? v[H + j] : 0.0;
    const double recv = __shfl_xor(up ? lo : hi, M);
    v[j] = (up ? hi : lo) + recv;
  }
  elem += up ? H : 0;
  mylen = up ? (mylen > H ? mylen - H : 0) : (mylen < H ? mylen : H);
}

__global__ __launch_bounds__(256) void marginalize_pairs_kernel(int nblocks, const double* __restrict__ poses,
                                                                const double* __restrict__ pts, const uint8_t* __restrict__ pfree,
                                                                const int32_t* __restrict__ opt, const double* __restrict__ oxy,
                                                                const double* __restrict__ oinfo, double huber,
                                                                const int32_t* __restrict__ bstart, const int32_t* __restrict__ bi,
                                                                const int32_t* __restrict__ bj, const int32_t* __restrict__ e_first,
                                                                const int32_t* __restrict__ e_second, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int blk = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blk >= nblocks) return;
  const int ci = bi[blk], cj = bj[blk];
  double acc[36];
#pragma unroll
  for (int t = 0; t < 36; ++t) acc[t] = 0.0;
  for (int e = bstart[blk] + lane; e < bstart[blk + 1]; e += 64) {
    const int k1 = e_first[e], k2 = e_second[e];  // the observations of the shared point in camera i and in camera j
    const int p = opt[k2];
    const int pf = pfree ? pfree[p] : 1;
    Obs o1, o2;
    const bool f1 = linearize<true>(poses + 7 * ci, GH_KF_SE3, pts + 3 * p, pf, oxy + 2 * k1, oinfo ? oinfo + 4 * k1 : nullptr, huber, o1);
    const bool f2 = linearize<true>(poses + 7 * cj, GH_KF_SE3, pts + 3 * p, pf, oxy + 2 * k2, oinfo ? oinfo + 4 * k2 : nullptr, huber, o2);
    if (!f1 || !f2) continue;  // behind one of the two cameras: no constraint
    double L1[4], L2[4];
    weighted_info(oinfo ? oinfo + 4 * k1 : nullptr, o1.w, L1);
    weighted_info(oinfo ? oinfo + 4 * k2 : nullptr, o2.w, L2);
    // LJc (2 x 6), LJp (2 x 3) of the observation in camera j
    double LJc[12], LJp2[6], LJp1[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      LJc[c] = L2[0] * o2.Jc[c] + L2[1] * o2.Jc[6 + c];
      LJc[6 + c] = L2[2] * o2.Jc[c] + L2[3] * o2.Jc[6 + c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      LJp2[c] = L2[0] * o2.Jp[c] + L2[1] * o2.Jp[3 + c];
      LJp2[3 + c] = L2[2] * o2.Jp[c] + L2[3] * o2.Jp[3 + c];
      LJp1[c] = L1[0] * o1.Jp[c] + L1[1] * o1.Jp[3 + c];
      LJp1[3 + c] = L1[2] * o1.Jp[c] + L1[3] * o1.Jp[3 + c];
    }
    double A[36], B[18], V[9];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int b = 0; b < 6; ++b) A[6 * a + b] = o2.Jc[a] * LJc[b] + o2.Jc[6 + a] * LJc[6 + b];
#pragma unroll
      for (int b = 0; b < 3; ++b) B[3 * a + b] = o2.Jc[a] * LJp2[b] + o2.Jc[6 + a] * LJp2[3 + b];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
        V[3 * a + b] = (o2.Jp[a] * LJp2[b] + o2.Jp[3 + a] * LJp2[3 + b]) + (o1.Jp[a] * LJp1[b] + o1.Jp[3 + a] * LJp1[3 + b]);
    // V^-1 by the adjugate (V symmetric positive definite for a point two cameras constrain)
    const double c00 = V[4] * V[8] - V[5] * V[7], c01 = V[5] * V[6] - V[3] * V[8], c02 = V[3] * V[7] - V[4] * V[6];
    const double det = V[0] * c00 + V[1] * c01 + V[2] * c02;
    if (pf && det > 0.0) {
      const double id = 1.0 / det;
      const double Vi[9] = {c00 * id, (V[2] * V[7] - V[1] * V[8]) * id, (V[1] * V[5] - V[2] * V[4]) * id,
                            c01 * id, (V[0] * V[8] - V[2] * V[6]) * id, (V[2] * V[3] - V[0] * V[5]) * id,
                            c02 * id, (V[1] * V[6] - V[0] * V[7]) * id, (V[0] * V[4] - V[1] * V[3]) * id};
      double BV[18];
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) BV[3 * a + b] = B[3 * a] * Vi[b] + B[3 * a + 1] * Vi[3 + b] + B[3 * a + 2] * Vi[6 + b];
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) A[6 * a + b] -= BV[3 * a] * B[3 * b] + BV[3 * a + 1] * B[3 * b + 1] + BV[3 * a + 2] * B[3 * b + 2];
    }
#pragma unroll
    for (int t = 0; t < 36; ++t) acc[t] += A[t];
  }
  int elem = 0, mylen = 36;
  scatter_step36<36, 32>(acc, lane, elem, mylen);
  scatter_step36<18, 16>(acc, lane, elem, mylen);
  scatter_step36<9, 8>(acc, lane, elem, mylen);
  scatter_step36<5, 4>(acc, lane, elem, mylen);
  scatter_step36<3, 2>(acc, lane, elem, mylen);
  scatter_step36<2, 1>(acc, lane, elem, mylen);
  if (mylen > 0) out[(size_t)36 * blk + elem] = acc[0];
}
}  // namespace

extern "C" gh_status gh_ba_marginalize(gh_ctx* ctx, const gh_ba_problem* pr, double huber_delta, int32_t min_shared,
                                       int32_t max_edges, int32_t* edge_first, int32_t* edge_second, int32_t* edge_shared,
                                       double* edge_info, int32_t* n_edges) {
  if (!ctx || !pr || !n_edges) return GH_ERR_ARG;
  GH_ENTER(ctx);
  const int nc = pr->n_cams, np = pr->n_points, no = pr->n_obs;
  GH_CHECK_ARG(ctx, nc >= 1 && np >= 0 && no >= 0 && max_edges >= 0 && min_shared >= 1);
  GH_CHECK_ARG(ctx, pr->cam_pose && (np == 0 || pr->point_xyz) && (no == 0 || (pr->obs_cam && pr->obs_point && pr->obs_xy)));
  GH_CHECK_ARG(ctx, max_edges == 0 || (edge_first && edge_second && edge_info));
  for (int k = 0; k < no; ++k)
    GH_CHECK_ARG(ctx, pr->obs_cam[k] >= 0 && pr->obs_cam[k] < nc && pr->obs_point[k] >= 0 && pr->obs_point[k] < np);
  // observations by point (original order inside a point)
  std::vector<int32_t> pstart((size_t)np + 1, 0), plist((size_t)no);
  for (int k = 0; k < no; ++k) ++pstart[pr->obs_point[k] + 1];
  for (int p = 0; p < np; ++p) pstart[p + 1] += pstart[p];
  {
    std::vector<int32_t> cur(pstart.begin(), pstart.end() - 1);
    for (int k = 0; k < no; ++k) plist[cur[pr->obs_point[k]]++] = k;
  }
  // the pairs of observations of one point in two different cameras, keyed by (lower camera, higher camera)
  struct Entry {
    int64_t key;
    int32_t k_first, k_second;
  };
  std::vector<Entry> ent;
  for (int p = 0; p < np; ++p)
    for (int a = pstart[p]; a < pstart[p + 1]; ++a)
      for (int b = a + 1; b < pstart[p + 1]; ++b) {
        const int ka = plist[a], kb = plist[b], ca = pr->obs_cam[ka], cb = pr->obs_cam[kb];
        if (ca == cb) continue;
        const int k1 = ca < cb ? ka : kb, k2 = ca < cb ? kb : ka;
        ent.push_back(Entry{(int64_t)std::min(ca, cb) * nc + std::max(ca, cb), k1, k2});
      }
  std::stable_sort(ent.begin(), ent.end(), [](const Entry& x, const Entry& y) { return x.key < y.key; });
  std::vector<int32_t> bstart, bi, bj, e1, e2;
  for (size_t b0 = 0; b0 < ent.size();) {
    size_t b1 = b0;
    while (b1 < ent.size() && ent[b1].key == ent[b0].key) ++b1;
    if ((int64_t)(b1 - b0) >= min_shared) {
      bstart.push_back((int32_t)e1.size());
      bi.push_back((int32_t)(ent[b0].key / nc));
      bj.push_back((int32_t)(ent[b0].key % nc));
      for (size_t e = b0; e < b1; ++e) {
        e1.push_back(ent[e].k_first);
        e2.push_back(ent[e].k_second);
      }
    }
    b0 = b1;
  }
  const int nblocks = (int)bi.size();
  bstart.push_back((int32_t)e1.size());
  *n_edges = nblocks;
  if (nblocks == 0 || max_edges == 0) return GH_OK;
  if (nblocks > max_edges)
    return gh_set_error(ctx, GH_ERR_ARG, "gh_ba_marginalize: %d camera pairs share >= %d points, room for %d", nblocks, min_shared, max_edges);
  // device image: one scratch block
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const bool has_info = pr->obs_info != nullptr, has_free = pr->point_free != nullptr;
  const size_t b_pose = pad((size_t)nc * 56), b_pts = pad((size_t)np * 24), b_free = pad(has_free ? (size_t)np : 0),
               b_opt = pad((size_t)no * 4), b_xy = pad((size_t)no * 16), b_inf = pad(has_info ? (size_t)no * 32 : 0),
               b_bs = pad(bstart.size() * 4), b_b = pad((size_t)nblocks * 4), b_e = pad(e1.size() * 4),
               b_out = pad((size_t)nblocks * 288);
  void* base = nullptr;
  GH_TRY(gh_scratch(ctx, b_pose + b_pts + b_free + b_opt + b_xy + b_inf + b_bs + 2 * b_b + 2 * b_e + b_out, &base));
  char* c = static_cast<char*>(base);
  auto take = [&](size_t b) { char* r = c; c += b; return r; };
  double* d_pose = (double*)take(b_pose);
  double* d_pts = (double*)take(b_pts);
  uint8_t* d_free = (uint8_t*)take(b_free);
  int32_t* d_opt = (int32_t*)take(b_opt);
  double* d_xy = (double*)take(b_xy);
  double* d_inf = (double*)take(b_inf);
  int32_t* d_bs = (int32_t*)take(b_bs);
  int32_t* d_bi = (int32_t*)take(b_b);
  int32_t* d_bj = (int32_t*)take(b_b);
  int32_t* d_e1 = (int32_t*)take(b_e);
  int32_t* d_e2 = (int32_t*)take(b_e);
  double* d_out = (double*)take(b_out);
  auto up = [&](void* dst, const void* src, size_t bytes) -> gh_status {
    if (bytes) GH_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return GH_OK;
  };
  GH_TRY(up(d_pose, pr->cam_pose, (size_t)nc * 56));
  GH_TRY(up(d_pts, pr->point_xyz, (size_t)np * 24));
  if (has_free) GH_TRY(up(d_free, pr->point_free, (size_t)np));
  GH_TRY(up(d_opt, pr->obs_point, (size_t)no * 4));
  GH_TRY(up(d_xy, pr->obs_xy, (size_t)no * 16));
  if (has_info) GH_TRY(up(d_inf, pr->obs_info, (size_t)no * 32));
  GH_TRY(up(d_bs, bstart.data(), bstart.size() * 4));
  GH_TRY(up(d_bi, bi.data(), (size_t)nblocks * 4));
  GH_TRY(up(d_bj, bj.data(), (size_t)nblocks * 4));
  GH_TRY(up(d_e1, e1.data(), e1.size() * 4));
  GH_TRY(up(d_e2, e2.data(), e2.size() * 4));
  GH_HIP(ctx, hipMemsetAsync(d_out, 0, (size_t)nblocks * 288, ctx->stream));
  GH_LAUNCH(ctx, "ba_marginalize", marginalize_pairs_kernel, dim3(gh_div_up(nblocks, 4)), dim3(256), 0, nblocks,
            (const double*)d_pose, (const double*)d_pts, has_free ? (const uint8_t*)d_free : nullptr, (const int32_t*)d_opt,
            (const double*)d_xy, has_info ? (const double*)d_inf : nullptr, huber_delta, (const int32_t*)d_bs,
            (const int32_t*)d_bi, (const int32_t*)d_bj, (const int32_t*)d_e1, (const int32_t*)d_e2, d_out);
  GH_HIP(ctx, hipMemcpyAsync(edge_info, d_out, (size_t)nblocks * 288, hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (the uploads above read the caller's and this function's host arrays)
  for (int b = 0; b < nblocks; ++b) {
    edge_first[b] = bi[b];
    edge_second[b] = bj[b];
    if (edge_shared) edge_shared[b] = bstart[b + 1] - bstart[b];
  }
  return GH_OK;
}
