// RANSAC model estimation with inlier masks on gfx950 — SURVEY.md 8 f3, the step right after matching.
//
// Fills in behind GSLAM::Estimator (GSLAM/core/Estimator.h:92-169: findHomography / findAffine2D / findFundamental /
// findAffine3D with `std::vector<uchar>* mask`).  The reference ships the INTERFACE only (the estimator plugin is
// commented out of the build, CMakeLists.txt:45), so the algorithm is specified here and restated in
// oracle/ransac_oracle.c; GPU and oracle agree bit for bit (models and masks):
//   * a fixed budget of 2048 hypotheses; hypothesis h draws its minimal sample with splitmix64(seed, h) (duplicates
//     rejected), solves the minimal problem in f64 by Gaussian elimination (no library calls, no FMA contraction),
//   * every hypothesis is scored against all N correspondences (squared error <= threshold^2), integer inlier counts,
//   * winner = most inliers, lowest hypothesis index on ties; its model and inlier mask are returned.
// Models: H (4 pairs, 8x8 system, h33 = 1, forward transfer error), A2 (3 pairs, 2x3 affine), F (8 pairs, Hartley-
// normalised 8x9 nullspace by full pivoting, Sampson error; rank 2 is not enforced), A3 (4 pairs, 3x4 affine, 3D).
// CDNA4 mapping: one lane per hypothesis for the tiny dense solves, one workgroup per hypothesis for scoring
// (coalesced point reads, integer block reduction), everything in one stream; the work is small and latency-bound.
#include "common.h"

namespace {

enum { kModelH = 0, kModelA2 = 1, kModelF = 2, kModelA3 = 3 };
constexpr int kHyp = 2048;
constexpr double kTiny = 1e-12;

__host__ __device__ inline uint64_t sm64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__device__ inline int sample_size(int model) { return model == kModelH ? 4 : (model == kModelA2 ? 3 : (model == kModelF ? 8 : 4)); }
__device__ inline int model_size(int model) { return model == kModelH ? 9 : (model == kModelA2 ? 6 : (model == kModelF ? 9 : 12)); }

// Solve A x = b (n <= 8, nrhs <= 3) in place, partial pivoting (first maximum).  a: n x (n + nrhs) row-major, ld = 12.
__device__ bool ge_solve(double (*a)[12], int n, int nrhs) {
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double best = fabs(a[k][k]);
    for (int r = k + 1; r < n; ++r) {
      const double v = fabs(a[r][k]);
      if (v > best) {
        best = v;
        piv = r;
      }
    }
    if (!(best > kTiny)) return false;
    if (piv != k)
      for (int c = 0; c < n + nrhs; ++c) {
        const double t = a[k][c];
        a[k][c] = a[piv][c];
        a[piv][c] = t;
      }
    const double inv = 1.0 / a[k][k];
    for (int r = k + 1; r < n; ++r) {
      const double f = a[r][k] * inv;
      for (int c = k; c < n + nrhs; ++c) a[r][c] = a[r][c] - f * a[k][c];
    }
  }
  for (int j = 0; j < nrhs; ++j)
    for (int r = n - 1; r >= 0; --r) {
      double s = a[r][n + j];
      for (int c = r + 1; c < n; ++c) s = s - a[r][c] * a[c][n + j];
      a[r][n + j] = s / a[r][r];
    }
  return true;
}

struct Norm {  // Hartley normalisation of both point sets (computed on the host over ALL points)
  double m1x, m1y, s1, m2x, m2y, s2;
};

// p: N x dim doubles (src), q: N x dim doubles (dst)
__global__ __launch_bounds__(64) void ransac_solve_kernel(int model, const double* __restrict__ p,
                                                          const double* __restrict__ q, int n, uint64_t seed, Norm nm,
                                                          double* __restrict__ models, int* __restrict__ valid) {
  const int h = blockIdx.x * 64 + threadIdx.x;
  if (h >= kHyp) return;
  const int s = sample_size(model);
  int idx[8];
  uint64_t st = sm64(seed ^ ((uint64_t)h * 0xD1B54A32D192ED03ull));
  for (int j = 0; j < s; ++j) {
    for (;;) {
      st = sm64(st);
      const int c = (int)(st % (uint64_t)n);
      bool dup = false;
      for (int t = 0; t < j; ++t) dup = dup || idx[t] == c;
      if (!dup) {
        idx[j] = c;
        break;
      }
    }
  }
  double a[8][12];
  double* out = models + (size_t)h * 12;
  bool ok = false;
  if (model == kModelH) {
    for (int j = 0; j < 4; ++j) {
      const double x = p[2 * idx[j]], y = p[2 * idx[j] + 1], u = q[2 * idx[j]], v = q[2 * idx[j] + 1];
      double* r0 = a[2 * j];
      double* r1 = a[2 * j + 1];
      r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -u * x; r0[7] = -u * y; r0[8] = u;
      r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -v * x; r1[7] = -v * y; r1[8] = v;
    }
    ok = ge_solve(a, 8, 1);
    if (ok) {
      for (int i = 0; i < 8; ++i) out[i] = a[i][8];
      out[8] = 1.0;
    }
  } else if (model == kModelA2) {
    for (int j = 0; j < 3; ++j) {
      a[j][0] = p[2 * idx[j]]; a[j][1] = p[2 * idx[j] + 1]; a[j][2] = 1;
      a[j][3] = q[2 * idx[j]]; a[j][4] = q[2 * idx[j] + 1];
    }
    ok = ge_solve(a, 3, 2);
    if (ok)
      for (int i = 0; i < 3; ++i) {
        out[i] = a[i][3];
        out[3 + i] = a[i][4];
      }
  } else if (model == kModelA3) {
    for (int j = 0; j < 4; ++j) {
      a[j][0] = p[3 * idx[j]]; a[j][1] = p[3 * idx[j] + 1]; a[j][2] = p[3 * idx[j] + 2]; a[j][3] = 1;
      a[j][4] = q[3 * idx[j]]; a[j][5] = q[3 * idx[j] + 1]; a[j][6] = q[3 * idx[j] + 2];
    }
    ok = ge_solve(a, 4, 3);
    if (ok)
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) out[4 * r + c] = a[c][4 + r];
  } else {  // fundamental: 8 x 9 nullspace with full pivoting
    for (int j = 0; j < 8; ++j) {
      const double x = (p[2 * idx[j]] - nm.m1x) * nm.s1, y = (p[2 * idx[j] + 1] - nm.m1y) * nm.s1;
      const double u = (q[2 * idx[j]] - nm.m2x) * nm.s2, v = (q[2 * idx[j] + 1] - nm.m2y) * nm.s2;
      double* r = a[j];
      r[0] = u * x; r[1] = u * y; r[2] = u; r[3] = v * x; r[4] = v * y; r[5] = v; r[6] = x; r[7] = y; r[8] = 1;
    }
    int perm[9];
    for (int c = 0; c < 9; ++c) perm[c] = c;
    ok = true;
    for (int k = 0; k < 8 && ok; ++k) {
      int pr = k, pc = k;
      double best = -1.0;
      for (int r = k; r < 8; ++r)
        for (int c = k; c < 9; ++c) {
          const double v = fabs(a[r][c]);
          if (v > best) {
            best = v;
            pr = r;
            pc = c;
          }
        }
      if (!(best > kTiny)) {
        ok = false;
        break;
      }
      if (pr != k)
        for (int c = 0; c < 9; ++c) {
          const double t = a[k][c];
          a[k][c] = a[pr][c];
          a[pr][c] = t;
        }
      if (pc != k) {
        for (int r = 0; r < 8; ++r) {
          const double t = a[r][k];
          a[r][k] = a[r][pc];
          a[r][pc] = t;
        }
        const int t = perm[k];
        perm[k] = perm[pc];
        perm[pc] = t;
      }
      const double inv = 1.0 / a[k][k];
      for (int r = k + 1; r < 8; ++r) {
        const double f = a[r][k] * inv;
        for (int c = k; c < 9; ++c) a[r][c] = a[r][c] - f * a[k][c];
      }
    }
    if (ok) {
      double z[9];
      z[8] = 1.0;
      for (int r = 7; r >= 0; --r) {
        double sres = 0.0;
        for (int c = r + 1; c < 9; ++c) sres = sres + a[r][c] * z[c];
        z[r] = -sres / a[r][r];
      }
      double fh[9];
      for (int c = 0; c < 9; ++c) fh[perm[c]] = z[c];
      // F = T2^T * Fh * T1, T = [s 0 -s m_x; 0 s -s m_y; 0 0 1]
      const double T1[9] = {nm.s1, 0, -nm.s1 * nm.m1x, 0, nm.s1, -nm.s1 * nm.m1y, 0, 0, 1};
      const double T2[9] = {nm.s2, 0, -nm.s2 * nm.m2x, 0, nm.s2, -nm.s2 * nm.m2y, 0, 0, 1};
      double tmp[9];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          double acc = 0.0;
          for (int k = 0; k < 3; ++k) acc = acc + fh[3 * r + k] * T1[3 * k + c];
          tmp[3 * r + c] = acc;
        }
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          double acc = 0.0;
          for (int k = 0; k < 3; ++k) acc = acc + T2[3 * k + r] * tmp[3 * k + c];
          out[3 * r + c] = acc;
        }
    }
  }
  valid[h] = ok ? 1 : 0;
}

// squared error of correspondence i under model m; returns false if undefined
__device__ inline bool model_error(int model, const double* m, const double* p, const double* q, int i, double* err) {
  if (model == kModelH) {
    const double x = p[2 * i], y = p[2 * i + 1];
    const double w = m[6] * x + m[7] * y + m[8];
    if (!(fabs(w) > kTiny)) return false;
    const double px = (m[0] * x + m[1] * y + m[2]) / w, py = (m[3] * x + m[4] * y + m[5]) / w;
    const double dx = px - q[2 * i], dy = py - q[2 * i + 1];
    *err = dx * dx + dy * dy;
    return true;
  }
  if (model == kModelA2) {
    const double x = p[2 * i], y = p[2 * i + 1];
    const double dx = (m[0] * x + m[1] * y + m[2]) - q[2 * i], dy = (m[3] * x + m[4] * y + m[5]) - q[2 * i + 1];
    *err = dx * dx + dy * dy;
    return true;
  }
  if (model == kModelA3) {
    const double X = p[3 * i], Y = p[3 * i + 1], Z = p[3 * i + 2];
    double e = 0.0;
    for (int r = 0; r < 3; ++r) {
      const double d = (m[4 * r] * X + m[4 * r + 1] * Y + m[4 * r + 2] * Z + m[4 * r + 3]) - q[3 * i + r];
      e = e + d * d;
    }
    *err = e;
    return true;
  }
  const double x = p[2 * i], y = p[2 * i + 1], u = q[2 * i], v = q[2 * i + 1];
  const double fx0 = m[0] * x + m[1] * y + m[2], fx1 = m[3] * x + m[4] * y + m[5], fx2 = m[6] * x + m[7] * y + m[8];
  const double ft0 = m[0] * u + m[3] * v + m[6], ft1 = m[1] * u + m[4] * v + m[7];
  const double num = u * fx0 + v * fx1 + fx2;
  const double den = fx0 * fx0 + fx1 * fx1 + ft0 * ft0 + ft1 * ft1;
  if (!(den > 1e-300)) return false;
  *err = (num * num) / den;
  return true;
}

__global__ __launch_bounds__(256) void ransac_score_kernel(int model, const double* __restrict__ p,
                                                           const double* __restrict__ q, int n, double thr2,
                                                           const double* __restrict__ models,
                                                           const int* __restrict__ valid, int* __restrict__ counts) {
  __shared__ int red[256];
  const int h = blockIdx.x;
  int c = 0;
  if (valid[h]) {
    double m[12];
    const int ms = model_size(model);
    for (int k = 0; k < ms; ++k) m[k] = models[(size_t)h * 12 + k];
    for (int i = threadIdx.x; i < n; i += 256) {
      double e;
      if (model_error(model, m, p, q, i, &e) && e <= thr2) ++c;
    }
  }
  red[threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[h] = valid[h] ? red[0] : -1;
}

__global__ __launch_bounds__(256) void ransac_best_kernel(const int* __restrict__ counts, int* __restrict__ best) {
  __shared__ long long red[256];
  long long key = -1;  // count << 32 | (kHyp - 1 - h): larger count wins, then lower index
  for (int h = threadIdx.x; h < kHyp; h += 256)
    if (counts[h] >= 0) {
      const long long k = ((long long)counts[h] << 32) | (long long)(kHyp - 1 - h);
      key = k > key ? k : key;
    }
  red[threadIdx.x] = key;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] = red[threadIdx.x + o] > red[threadIdx.x] ? red[threadIdx.x + o] : red[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    best[0] = red[0] < 0 ? -1 : kHyp - 1 - (int)(red[0] & 0xFFFFFFFFll);
    best[1] = red[0] < 0 ? 0 : (int)(red[0] >> 32);
  }
}

__global__ __launch_bounds__(256) void ransac_mask_kernel(int model, const double* __restrict__ p,
                                                          const double* __restrict__ q, int n, double thr2,
                                                          const double* __restrict__ models,
                                                          const int* __restrict__ best, uint8_t* __restrict__ mask,
                                                          double* __restrict__ model_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int h = best[0];
  if (h < 0) {
    if (i < n) mask[i] = 0;
    return;
  }
  double m[12];
  const int ms = model_size(model);
  for (int k = 0; k < ms; ++k) m[k] = models[(size_t)h * 12 + k];
  if (i < 12) model_out[i] = i < ms ? m[i] : 0.0;
  if (i < n) {
    double e;
    mask[i] = (model_error(model, m, p, q, i, &e) && e <= thr2) ? 1 : 0;
  }
}

}  // namespace

extern "C" gh_status gh_ransac_estimate(gh_ctx* ctx, int model, const double* src, const double* dst, int n,
                                        double threshold, uint64_t seed, double* model_out, uint8_t* mask_out,
                                        int* inliers_out) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, model >= 0 && model <= 3 && src && dst && model_out && inliers_out && threshold >= 0);
  const int dim = model == kModelA3 ? 3 : 2;
  const int s = model == kModelH ? 4 : (model == kModelA2 ? 3 : (model == kModelF ? 8 : 4));
  *inliers_out = 0;
  for (int k = 0; k < 12; ++k) model_out[k] = 0.0;
  if (mask_out)
    for (int i = 0; i < n; ++i) mask_out[i] = 0;
  if (n < s) return GH_OK;  // not enough correspondences: no model (inliers 0)
  GH_HIP(ctx, hipSetDevice(ctx->device));
  Norm nm = {0, 0, 1, 0, 0, 1};
  if (model == kModelF) {  // Hartley normalisation, sequential sums in index order (the oracle does the same)
    double ax = 0, ay = 0, bx = 0, by = 0;
    for (int i = 0; i < n; ++i) {
      ax += src[2 * i]; ay += src[2 * i + 1];
      bx += dst[2 * i]; by += dst[2 * i + 1];
    }
    nm.m1x = ax / n; nm.m1y = ay / n; nm.m2x = bx / n; nm.m2y = by / n;
    double d1 = 0, d2 = 0;
    for (int i = 0; i < n; ++i) {
      const double x = src[2 * i] - nm.m1x, y = src[2 * i + 1] - nm.m1y;
      const double u = dst[2 * i] - nm.m2x, v = dst[2 * i + 1] - nm.m2y;
      d1 += sqrt(x * x + y * y);
      d2 += sqrt(u * u + v * v);
    }
    d1 /= n; d2 /= n;
    nm.s1 = d1 > 0 ? 1.4142135623730951 / d1 : 1.0;
    nm.s2 = d2 > 0 ? 1.4142135623730951 / d2 : 1.0;
  }
  const size_t pb = (((size_t)n * dim * 8) + 255) & ~(size_t)255;
  const size_t off_q = pb, off_models = 2 * pb, off_valid = off_models + (size_t)kHyp * 12 * 8,
               off_counts = off_valid + kHyp * 4, off_best = off_counts + kHyp * 4, off_mout = off_best + 256,
               off_mask = off_mout + 256, total = off_mask + (((size_t)n + 255) & ~(size_t)255);
  void* base = nullptr;
  GH_TRY(gh_scratch(ctx, total, &base));
  uint8_t* b = (uint8_t*)base;
  double* d_p = (double*)b;
  double* d_q = (double*)(b + off_q);
  double* d_models = (double*)(b + off_models);
  int* d_valid = (int*)(b + off_valid);
  int* d_counts = (int*)(b + off_counts);
  int* d_best = (int*)(b + off_best);
  double* d_mout = (double*)(b + off_mout);
  uint8_t* d_mask = b + off_mask;
  GH_HIP(ctx, hipMemcpyAsync(d_p, src, (size_t)n * dim * 8, hipMemcpyHostToDevice, ctx->stream));
  GH_HIP(ctx, hipMemcpyAsync(d_q, dst, (size_t)n * dim * 8, hipMemcpyHostToDevice, ctx->stream));
  const double thr2 = threshold * threshold;
  GH_LAUNCH(ctx, "ransac_solve", ransac_solve_kernel, dim3(kHyp / 64), dim3(64), 0, model, d_p, d_q, n, seed, nm,
            d_models, d_valid);
  GH_LAUNCH(ctx, "ransac_score", ransac_score_kernel, dim3(kHyp), dim3(256), 0, model, d_p, d_q, n, thr2, d_models,
            d_valid, d_counts);
  GH_LAUNCH(ctx, "ransac_best", ransac_best_kernel, dim3(1), dim3(256), 0, d_counts, d_best);
  GH_LAUNCH(ctx, "ransac_mask", ransac_mask_kernel, dim3(gh_div_up(n > 12 ? n : 12, 256)), dim3(256), 0, model, d_p, d_q,
            n, thr2, d_models, d_best, d_mask, d_mout);
  int best[2] = {-1, 0};
  GH_HIP(ctx, hipMemcpyAsync(best, d_best, 8, hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(ctx, hipMemcpyAsync(model_out, d_mout, 12 * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (mask_out) GH_HIP(ctx, hipMemcpyAsync(mask_out, d_mask, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (best[0] < 0) {
    for (int k = 0; k < 12; ++k) model_out[k] = 0.0;
    return GH_OK;
  }
  *inliers_out = best[1];
  return GH_OK;
}
