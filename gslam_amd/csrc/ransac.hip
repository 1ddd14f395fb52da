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

enum { kModelH = 0, kModelA2 = 1, kModelF = 2, kModelA3 = 3, kModelE = 4, kModelSim3 = 5, kModelPlane = 6, kModelPnP = 7 };
constexpr int kHyp = 2048;
constexpr double kTiny = 1e-12;

__host__ __device__ inline uint64_t sm64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__host__ __device__ inline int sample_size(int model) {
  switch (model) {
    case kModelH: return 4;
    case kModelA2: return 3;
    case kModelF: case kModelE: return 8;
    case kModelA3: return 4;
    case kModelSim3: case kModelPlane: return 3;
    default: return 6;  // PnP: 6-point DLT
  }
}
__host__ __device__ inline int model_size(int model) {
  switch (model) {
    case kModelH: case kModelF: case kModelE: return 9;
    case kModelA2: return 6;
    case kModelSim3: return 8;
    case kModelPlane: return 4;
    default: return 12;  // A3 (3 x 4) and PnP ([R | t], world -> camera)
  }
}
__host__ __device__ inline int dim_p(int model) { return (model == kModelA3 || model == kModelSim3 || model == kModelPlane || model == kModelPnP) ? 3 : 2; }
__host__ __device__ inline int dim_q(int model) { return (model == kModelA3 || model == kModelSim3 || model == kModelPlane) ? 3 : 2; }

// Cyclic Jacobi eigen-decomposition of a symmetric N x N matrix (a is destroyed, v receives the eigenvectors as columns):
// a fixed number of sweeps of the classical rotation, written with + - * / sqrt only, so that the device and the CPU
// checker produce the same bits.
template <int N>
__host__ __device__ inline void jacobi_eig(double (*a)[N], double (*v)[N]) {
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) v[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; ++sweep)
    for (int p = 0; p < N - 1; ++p)
      for (int q = p + 1; q < N; ++q) {
        const double apq = a[p][q];
        if (!(fabs(apq) > 1e-300)) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < N; ++k) {  // A <- A J
          const double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - sn * akq;
          a[k][q] = sn * akp + c * akq;
        }
        for (int k = 0; k < N; ++k) {  // A <- J^T A
          const double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - sn * aqk;
          a[q][k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < N; ++k) {
          const double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = c * vkp - sn * vkq;
          v[k][q] = sn * vkp + c * vkq;
        }
      }
}

// Horn's closed-form absolute orientation with scale (Horn 1987; GSLAM::Estimator::findSIM3, method S3_Horn) from three
// point pairs: b ~ s R a + t.  out = [qx qy qz qw tx ty tz s] (GSLAM's SIM3 field order).
// (m pairs: the three of a RANSAC sample through idx, or all n in index order with idx = nullptr -- the NOSAMPLE fit)
__host__ __device__ inline bool solve_sim3(const double* p, const double* q, const int* idx, double* out, int m = 3) {
  double ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0};
  for (int j = 0; j < m; ++j) {
    const int ij = idx ? idx[j] : j;
    for (int e = 0; e < 3; ++e) {
      ca[e] = ca[e] + p[3 * ij + e];
      cb[e] = cb[e] + q[3 * ij + e];
    }
  }
  for (int e = 0; e < 3; ++e) {
    ca[e] = ca[e] / (double)m;
    cb[e] = cb[e] / (double)m;
  }
  double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, na = 0.0, nb = 0.0;
  for (int j = 0; j < m; ++j) {
    const int ij = idx ? idx[j] : j;
    double a[3], b[3];
    for (int e = 0; e < 3; ++e) {
      a[e] = p[3 * ij + e] - ca[e];
      b[e] = q[3 * ij + e] - cb[e];
      na = na + a[e] * a[e];
      nb = nb + b[e] * b[e];
    }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) S[r][c] = S[r][c] + a[r] * b[c];
  }
  if (!(na > kTiny) || !(nb > kTiny)) return false;
  double N[4][4] = {{S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0]},
                    {0, S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2]},
                    {0, 0, -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1]},
                    {0, 0, 0, -S[0][0] - S[1][1] + S[2][2]}};
  for (int r = 1; r < 4; ++r)
    for (int c = 0; c < r; ++c) N[r][c] = N[c][r];
  double V[4][4];
  jacobi_eig<4>(N, V);
  int best = 0;
  for (int k = 1; k < 4; ++k)
    if (N[k][k] > N[best][best]) best = k;
  double qw = V[0][best], qx = V[1][best], qy = V[2][best], qz = V[3][best];
  const double qn = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
  if (!(qn > kTiny)) return false;
  if (qw < 0) { qw = -qw; qx = -qx; qy = -qy; qz = -qz; }
  qw = qw / qn; qx = qx / qn; qy = qy / qn; qz = qz / qn;
  const double sc = sqrt(nb / na);
  const double R[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy),
                       2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx),
                       2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)};
  out[0] = qx; out[1] = qy; out[2] = qz; out[3] = qw;
  for (int r = 0; r < 3; ++r) out[4 + r] = cb[r] - sc * (R[3 * r] * ca[0] + R[3 * r + 1] * ca[1] + R[3 * r + 2] * ca[2]);
  out[7] = sc;
  return true;
}

// Plane through three points: out = [nx ny nz d], n unit, n . x + d = 0.
__device__ inline bool solve_plane(const double* p, const int* idx, double* out) {
  const double* p0 = p + 3 * idx[0];
  const double* p1 = p + 3 * idx[1];
  const double* p2 = p + 3 * idx[2];
  const double u[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, v[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
  double n[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
  const double len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  if (!(len > kTiny)) return false;
  for (int e = 0; e < 3; ++e) n[e] = n[e] / len;
  out[0] = n[0]; out[1] = n[1]; out[2] = n[2];
  out[3] = -(n[0] * p0[0] + n[1] * p0[1] + n[2] * p0[2]);
  return true;
}

// Perspective-n-point from six 3D-2D pairs by the direct linear transform: 12 x 12 homogeneous system, nullspace by
// elimination with full pivoting (as for F), scaled to |r3| = 1 with positive depth, rotation made orthonormal by
// Gram-Schmidt on its rows.  out = [R (row-major 9) | t (3)], X_c = R X_w + t.  Coplanar object points are degenerate.
__host__ __device__ inline bool pnp_from_projection(double* P, const double* X0, double* out);
__device__ inline bool solve_pnp_dlt(const double* p, const double* q, const int* idx, double* out) {
  double a[12][12];
  for (int j = 0; j < 6; ++j) {
    const double X = p[3 * idx[j]], Y = p[3 * idx[j] + 1], Z = p[3 * idx[j] + 2], u = q[2 * idx[j]], v = q[2 * idx[j] + 1];
    double* r0 = a[2 * j];
    double* r1 = a[2 * j + 1];
    r0[0] = X; r0[1] = Y; r0[2] = Z; r0[3] = 1; r0[4] = 0; r0[5] = 0; r0[6] = 0; r0[7] = 0;
    r0[8] = -u * X; r0[9] = -u * Y; r0[10] = -u * Z; r0[11] = -u;
    r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = 0; r1[4] = X; r1[5] = Y; r1[6] = Z; r1[7] = 1;
    r1[8] = -v * X; r1[9] = -v * Y; r1[10] = -v * Z; r1[11] = -v;
  }
  int perm[12];
  for (int c = 0; c < 12; ++c) perm[c] = c;
  for (int k = 0; k < 11; ++k) {
    int pr = k, pc = k;
    double best = -1.0;
    for (int r = k; r < 12; ++r)
      for (int c = k; c < 12; ++c) {
        const double v = fabs(a[r][c]);
        if (v > best) {
          best = v;
          pr = r;
          pc = c;
        }
      }
    if (!(best > kTiny)) return false;
    if (pr != k)
      for (int c = 0; c < 12; ++c) {
        const double t = a[k][c];
        a[k][c] = a[pr][c];
        a[pr][c] = t;
      }
    if (pc != k) {
      for (int r = 0; r < 12; ++r) {
        const double t = a[r][k];
        a[r][k] = a[r][pc];
        a[r][pc] = t;
      }
      const int t = perm[k];
      perm[k] = perm[pc];
      perm[pc] = t;
    }
    const double inv = 1.0 / a[k][k];
    for (int r = k + 1; r < 12; ++r) {
      const double f = a[r][k] * inv;
      for (int c = k; c < 12; ++c) a[r][c] = a[r][c] - f * a[k][c];
    }
  }
  double z[12], P[12];
  z[11] = 1.0;
  for (int r = 10; r >= 0; --r) {
    double sres = 0.0;
    for (int c = r + 1; c < 12; ++c) sres = sres + a[r][c] * z[c];
    z[r] = -sres / a[r][r];
  }
  for (int c = 0; c < 12; ++c) P[perm[c]] = z[c];
  return pnp_from_projection(P, p + 3 * idx[0], out);
}

// [R | t] from a 3 x 4 projection known up to scale: scaled to |r3| = 1 with X0 in front of the camera, rotation made
// orthonormal by Gram-Schmidt on its rows.
__host__ __device__ inline bool pnp_from_projection(double* P, const double* X0, double* out) {
  const double n3 = sqrt(P[8] * P[8] + P[9] * P[9] + P[10] * P[10]);
  if (!(n3 > kTiny)) return false;
  double lam = 1.0 / n3;
  if ((P[8] * X0[0] + P[9] * X0[1] + P[10] * X0[2] + P[11]) * lam < 0) lam = -lam;  // the sample lies in front of the camera
  for (int c = 0; c < 12; ++c) P[c] = P[c] * lam;
  double r1[3] = {P[0], P[1], P[2]}, r2[3] = {P[4], P[5], P[6]};
  const double n1 = sqrt(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]);
  if (!(n1 > kTiny)) return false;
  for (int e = 0; e < 3; ++e) r1[e] = r1[e] / n1;
  const double d12 = r2[0] * r1[0] + r2[1] * r1[1] + r2[2] * r1[2];
  for (int e = 0; e < 3; ++e) r2[e] = r2[e] - d12 * r1[e];
  const double n2 = sqrt(r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2]);
  if (!(n2 > kTiny)) return false;
  for (int e = 0; e < 3; ++e) r2[e] = r2[e] / n2;
  const double r3[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
  if (!(r3[0] * P[8] + r3[1] * P[9] + r3[2] * P[10] > 0)) return false;  // a reflection, not a rotation
  for (int e = 0; e < 3; ++e) {
    out[e] = r1[e];
    out[3 + e] = r2[e];
    out[6 + e] = r3[e];
  }
  out[9] = P[3] / n1;  // each translation component carries the scale of its own row of the raw estimate
  out[10] = P[7] / n2;
  out[11] = P[11];
  return true;
}

// Solve A x = b (n <= 8, nrhs <= 3) in place, partial pivoting (first maximum).  a: n x (n + nrhs) row-major, ld = 12.
__host__ __device__ inline bool ge_solve(double (*a)[12], int n, int nrhs) {
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
  } else if (model == kModelSim3) {
    ok = solve_sim3(p, q, idx, out);
  } else if (model == kModelPlane) {
    ok = solve_plane(p, idx, out);
  } else if (model == kModelPnP) {
    ok = solve_pnp_dlt(p, q, idx, out);
  } else {  // fundamental / essential: 8 x 9 nullspace with full pivoting
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
  if (model == kModelSim3) {
    const double qx = m[0], qy = m[1], qz = m[2], qw = m[3], sc = m[7];
    const double R[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy),
                         2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx),
                         2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)};
    const double X = p[3 * i], Y = p[3 * i + 1], Z = p[3 * i + 2];
    double e = 0.0;
    for (int r = 0; r < 3; ++r) {
      const double d = (sc * (R[3 * r] * X + R[3 * r + 1] * Y + R[3 * r + 2] * Z) + m[4 + r]) - q[3 * i + r];
      e = e + d * d;
    }
    *err = e;
    return true;
  }
  if (model == kModelPlane) {
    const double d = m[0] * p[3 * i] + m[1] * p[3 * i + 1] + m[2] * p[3 * i + 2] + m[3];
    *err = d * d;
    return true;
  }
  if (model == kModelPnP) {
    const double X = p[3 * i], Y = p[3 * i + 1], Z = p[3 * i + 2];
    const double zc = m[6] * X + m[7] * Y + m[8] * Z + m[11];
    if (!(zc > kTiny)) return false;
    const double dx = (m[0] * X + m[1] * Y + m[2] * Z + m[9]) / zc - q[2 * i];
    const double dy = (m[3] * X + m[4] * Y + m[5] * Z + m[10]) / zc - q[2 * i + 1];
    *err = dx * dx + dy * dy;
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

// Projection of the winning 8-point estimate onto the essential manifold (two equal singular values, one zero):
// E = U diag(s, s, 0) V^T with s = (s1 + s2) / 2, through the Jacobi eigen-decomposition of E^T E.  Host side, once.
bool project_essential(double* E) {
  double B[3][3], V[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) acc = acc + E[3 * k + r] * E[3 * k + c];
      B[r][c] = acc;
    }
  jacobi_eig<3>(B, V);
  int o[3] = {0, 1, 2};  // eigenvalues in descending order (stable selection)
  for (int a = 0; a < 2; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (B[o[b]][o[b]] > B[o[a]][o[a]]) {
        const int t = o[a];
        o[a] = o[b];
        o[b] = t;
      }
  const double l1 = B[o[0]][o[0]], l2 = B[o[1]][o[1]];
  if (!(l2 > 1e-300)) return false;
  const double s1 = sqrt(l1), s2 = sqrt(l2), sm = (s1 + s2) / 2.0;
  double u[2][3];
  for (int a = 0; a < 2; ++a) {
    const double sv = a == 0 ? s1 : s2;
    for (int r = 0; r < 3; ++r) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) acc = acc + E[3 * r + k] * V[k][o[a]];
      u[a][r] = acc / sv;
    }
  }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) E[3 * r + c] = sm * (u[0][r] * V[c][o[0]] + u[1][r] * V[c][o[1]]);
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

// LMedS score of a hypothesis: the element of rank n / 2 (ascending, 0-based) of its squared errors; a correspondence whose
// error is undefined counts as +inf.  Squared errors are non-negative doubles, whose bit patterns order as the values do:
// an exact radix select, 8 bits per pass, one workgroup per hypothesis (the errors are recomputed in every pass: ~50 flops each).
__global__ __launch_bounds__(256) void ransac_median_kernel(int model, const double* __restrict__ p,
                                                            const double* __restrict__ q, int n,
                                                            const double* __restrict__ models,
                                                            const int* __restrict__ valid,
                                                            unsigned long long* __restrict__ med) {
  __shared__ unsigned hist[256];
  __shared__ unsigned long long s_prefix;
  __shared__ unsigned s_rank;
  const int h = blockIdx.x, tid = threadIdx.x;
  if (!valid[h]) {
    if (tid == 0) med[h] = ~0ull;
    return;
  }
  double m[12];
  const int ms = model_size(model);
  for (int k = 0; k < ms; ++k) m[k] = models[(size_t)h * 12 + k];
  unsigned long long prefix = 0ull;
  unsigned rank = (unsigned)(n / 2);
  for (int pass = 7; pass >= 0; --pass) {
    hist[tid] = 0u;
    __syncthreads();
    const unsigned long long hi_mask = pass == 7 ? 0ull : (~0ull << (8 * (pass + 1)));
    for (int i = tid; i < n; i += 256) {
      double e;
      const unsigned long long b = (model_error(model, m, p, q, i, &e) && e == e) ? (unsigned long long)__double_as_longlong(e)
                                                                                 : 0x7FF0000000000000ull;
      if ((b & hi_mask) == prefix) atomicAdd(&hist[(unsigned)(b >> (8 * pass)) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned acc = 0u;
      int d = 0;
      for (; d < 255; ++d) {
        if (acc + hist[d] > rank) break;
        acc += hist[d];
      }
      s_prefix = prefix | ((unsigned long long)d << (8 * pass));
      s_rank = rank - acc;
    }
    __syncthreads();
    prefix = s_prefix;
    rank = s_rank;
    __syncthreads();
  }
  if (tid == 0) med[h] = prefix;
}

// NOSAMPLE (GSLAM/core/Estimator.h:89): the model from ALL correspondences, no hypotheses -- the algebraic least-squares
// counterpart of each minimal solver, sums taken sequentially in index order (host side: n x a few dozen flops; the oracle
// adds in the same order).  H, A2, A3: normal equations of the minimal solver's rows, Gaussian elimination.  F / E, PnP:
// eigenvector of the smallest eigenvalue of A^T A (cyclic Jacobi), then the minimal solver's own post-processing.  SIM3:
// Horn's closed form over all pairs.  Plane: normal = eigenvector of the smallest eigenvalue of the covariance.
template <int N>
static int smallest_eig_vector(double (*A)[N], double* vec) {
  double V[N][N];
  jacobi_eig<N>(A, V);
  int best = 0;
  for (int k = 1; k < N; ++k)
    if (A[k][k] < A[best][best]) best = k;
  for (int k = 0; k < N; ++k) vec[k] = V[k][best];
  return best;
}

static bool fit_all(int model, const double* p, const double* q, int n, const Norm& nm, double* out) {
  for (int k = 0; k < 12; ++k) out[k] = 0.0;
  if (model == kModelH || model == kModelA2 || model == kModelA3) {
    const int nu = model == kModelH ? 8 : (model == kModelA2 ? 3 : 4), nr = model == kModelH ? 1 : (model == kModelA2 ? 2 : 3);
    double N[8][12];
    for (int r = 0; r < 8; ++r)
      for (int c = 0; c < 12; ++c) N[r][c] = 0.0;
    auto add_row = [&](const double* r) {
      for (int a = 0; a < nu; ++a)
        for (int b = 0; b < nu + nr; ++b) N[a][b] = N[a][b] + r[a] * r[b];
    };
    for (int i = 0; i < n; ++i) {
      if (model == kModelH) {
        const double x = p[2 * i], y = p[2 * i + 1], u = q[2 * i], v = q[2 * i + 1];
        const double r0[9] = {x, y, 1, 0, 0, 0, -u * x, -u * y, u}, r1[9] = {0, 0, 0, x, y, 1, -v * x, -v * y, v};
        add_row(r0);
        add_row(r1);
      } else if (model == kModelA2) {
        const double r[5] = {p[2 * i], p[2 * i + 1], 1, q[2 * i], q[2 * i + 1]};
        add_row(r);
      } else {
        const double r[7] = {p[3 * i], p[3 * i + 1], p[3 * i + 2], 1, q[3 * i], q[3 * i + 1], q[3 * i + 2]};
        add_row(r);
      }
    }
    if (!ge_solve(N, nu, nr)) return false;
    if (model == kModelH) {
      for (int k = 0; k < 8; ++k) out[k] = N[k][8];
      out[8] = 1.0;
    } else if (model == kModelA2) {
      for (int k = 0; k < 3; ++k) {
        out[k] = N[k][3];
        out[3 + k] = N[k][4];
      }
    } else {
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) out[4 * r + c] = N[c][4 + r];
    }
    return true;
  }
  if (model == kModelSim3) return solve_sim3(p, q, nullptr, out, n);
  if (model == kModelPlane) {
    double c[3] = {0, 0, 0}, C[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, nv[3];
    for (int i = 0; i < n; ++i)
      for (int e = 0; e < 3; ++e) c[e] = c[e] + p[3 * i + e];
    for (int e = 0; e < 3; ++e) c[e] = c[e] / (double)n;
    for (int i = 0; i < n; ++i) {
      const double d[3] = {p[3 * i] - c[0], p[3 * i + 1] - c[1], p[3 * i + 2] - c[2]};
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) C[a][b] = C[a][b] + d[a] * d[b];
    }
    smallest_eig_vector<3>(C, nv);
    const double len = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
    if (!(len > kTiny)) return false;
    for (int e = 0; e < 3; ++e) out[e] = nv[e] / len;
    out[3] = -(out[0] * c[0] + out[1] * c[1] + out[2] * c[2]);
    return true;
  }
  if (model == kModelPnP) {
    double A[12][12];
    for (int r = 0; r < 12; ++r)
      for (int c = 0; c < 12; ++c) A[r][c] = 0.0;
    for (int i = 0; i < n; ++i) {
      const double X = p[3 * i], Y = p[3 * i + 1], Z = p[3 * i + 2], u = q[2 * i], v = q[2 * i + 1];
      const double r0[12] = {X, Y, Z, 1, 0, 0, 0, 0, -u * X, -u * Y, -u * Z, -u}, r1[12] = {0, 0, 0, 0, X, Y, Z, 1, -v * X, -v * Y, -v * Z, -v};
      for (int a = 0; a < 12; ++a)
        for (int b = 0; b < 12; ++b) A[a][b] = A[a][b] + (r0[a] * r0[b] + r1[a] * r1[b]);
    }
    double P[12];
    smallest_eig_vector<12>(A, P);
    return pnp_from_projection(P, p, out);
  }
  // fundamental / essential
  double A[9][9];
  for (int r = 0; r < 9; ++r)
    for (int c = 0; c < 9; ++c) A[r][c] = 0.0;
  for (int i = 0; i < n; ++i) {
    const double x = (p[2 * i] - nm.m1x) * nm.s1, y = (p[2 * i + 1] - nm.m1y) * nm.s1;
    const double u = (q[2 * i] - nm.m2x) * nm.s2, v = (q[2 * i + 1] - nm.m2y) * nm.s2;
    const double r[9] = {u * x, u * y, u, v * x, v * y, v, x, y, 1};
    for (int a = 0; a < 9; ++a)
      for (int b = 0; b < 9; ++b) A[a][b] = A[a][b] + r[a] * r[b];
  }
  double fh[9];
  smallest_eig_vector<9>(A, fh);
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
  return true;
}

}  // namespace

// The adaptive stopping rule of sequential RANSAC (Fischler & Bolles; the rule behind a `confidence` argument): walking the
// hypotheses in index order, after every strict improvement of the best inlier count c the number of hypotheses needed is
// N = ceil(log(1 - confidence) / log(1 - (c / n)^s)), and the walk stops once h + 1 >= N.  All kHyp hypotheses are scored
// in parallel anyway; the rule only decides WHICH prefix of them a sequential run would have looked at, so the result is
// the one a CPU implementation with the same draws returns.  Evaluated on the host with libm (the oracle makes the same
// calls), never on the device.  confidence outside (0, 1): every hypothesis counts.
static int adaptive_prefix_best(const int* counts, int n, int s, double confidence, int* best_count, int* used) {
  int best_h = -1, best_c = -1, limit = kHyp;
  const bool adaptive = confidence > 0.0 && confidence < 1.0;
  int h = 0;
  for (; h < limit; ++h) {
    if (counts[h] > best_c) {
      best_c = counts[h];
      best_h = h;
      if (adaptive && best_c > 0) {
        const double pg = pow((double)best_c / (double)n, (double)s);
        int need = kHyp;
        if (pg >= 1.0) need = h + 1;
        else if (pg > 0.0) {
          const double v = ceil(log(1.0 - confidence) / log(1.0 - pg));
          need = v < 1.0 ? 1 : (v > (double)kHyp ? kHyp : (int)v);
        }
        if (need < limit) limit = need < h + 1 ? h + 1 : need;
      }
    }
  }
  *best_count = best_c < 0 ? 0 : best_c;
  *used = h;
  return best_c < 0 ? -1 : best_h;
}

extern "C" gh_status gh_ransac_estimate(gh_ctx* ctx, int model, const double* src, const double* dst, int n,
                                        double threshold, uint64_t seed, double* model_out, uint8_t* mask_out,
                                        int* inliers_out) {
  return gh_ransac_estimate_conf(ctx, model, src, dst, n, threshold, 1.0, seed, model_out, mask_out, inliers_out, nullptr);
}

extern "C" gh_status gh_ransac_estimate_conf(gh_ctx* ctx, int model, const double* src, const double* dst, int n,
                                             double threshold, double confidence, uint64_t seed, double* model_out,
                                             uint8_t* mask_out, int* inliers_out, int* hypotheses_used_out) {
  return gh_ransac_estimate_ex(ctx, model, src, dst, n, threshold, confidence, seed, GH_SAMPLE_RANSAC, model_out, mask_out,
                               inliers_out, hypotheses_used_out);
}

extern "C" gh_status gh_ransac_estimate_ex(gh_ctx* ctx, int model, const double* src, const double* dst, int n,
                                           double threshold, double confidence, uint64_t seed, int sampling, double* model_out,
                                           uint8_t* mask_out, int* inliers_out, int* hypotheses_used_out) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, model >= 0 && model <= 7 && src && dst && model_out && inliers_out && threshold >= 0);
  GH_CHECK_ARG(ctx, sampling == GH_SAMPLE_RANSAC || sampling == GH_SAMPLE_LMEDS || sampling == GH_SAMPLE_NONE);
  const int dim = dim_p(model), dimq = dim_q(model);
  const int s = sample_size(model);
  *inliers_out = 0;
  if (hypotheses_used_out) *hypotheses_used_out = 0;
  for (int k = 0; k < 12; ++k) model_out[k] = 0.0;
  if (mask_out)
    for (int i = 0; i < n; ++i) mask_out[i] = 0;
  if (n < s) return GH_OK;  // not enough correspondences: no model (inliers 0)
  GH_HIP(ctx, hipSetDevice(ctx->device));
  Norm nm = {0, 0, 1, 0, 0, 1};
  if (model == kModelF || model == kModelE) {  // Hartley normalisation, sequential sums in index order (the oracle does the same)
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
  const size_t pb = (((size_t)n * 3 * 8) + 255) & ~(size_t)255;  // sized for the wider of the two point sets
  // (the counts slot holds 8 bytes per hypothesis: LMedS stores its medians there as 64-bit keys)
  const size_t off_q = pb, off_models = 2 * pb, off_valid = off_models + (size_t)kHyp * 12 * 8,
               off_counts = off_valid + kHyp * 4, off_best = off_counts + kHyp * 8, off_mout = off_best + 256,
               off_mask = off_mout + 256, total = off_mask + (((size_t)n + 255) & ~(size_t)255);
  void *base = nullptr, *hbase = nullptr;
  GH_TRY(gh_scratch(ctx, total, &base));
  GH_TRY(gh_pinned(ctx, total, &hbase));  // host mirror of the same layout: one DMA per direction and phase
  uint8_t *b = (uint8_t*)base, *hb = (uint8_t*)hbase;
  double* d_p = (double*)b;
  double* d_q = (double*)(b + off_q);
  double* d_models = (double*)(b + off_models);
  int* d_valid = (int*)(b + off_valid);
  int* d_counts = (int*)(b + off_counts);
  int* d_best = (int*)(b + off_best);
  double* d_mout = (double*)(b + off_mout);
  uint8_t* d_mask = b + off_mask;
  // (a per-frame caller -- Estimator::findFundamental / findHomography / findPnPRansac -- paid two pageable uploads and
  // three to five pageable downloads of ~30-50 us each around ~100 us of kernels)
  memcpy(hb, src, (size_t)n * dim * 8);
  memcpy(hb + off_q, dst, (size_t)n * dimq * 8);
  GH_HIP(ctx, hipMemcpyAsync(b, hb, off_q + (size_t)n * dimq * 8, hipMemcpyHostToDevice, ctx->stream));
  double thr2 = threshold * threshold;
  int best[2] = {-1, 0};
  if (sampling == GH_SAMPLE_NONE) {
    // the all-point fit is hypothesis 0; the device evaluates it against every correspondence (mask + inlier count)
    double m[12];
    const bool ok = fit_all(model, src, dst, n, nm, m);
    if (hypotheses_used_out) *hypotheses_used_out = ok ? 1 : 0;
    if (!ok) {  // degenerate fit: outputs stay zero (cleared at entry); the upload from the pinned block must not outlive the call
      GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
      return GH_OK;
    }
    memcpy(hb + off_models, m, sizeof(m));
    best[0] = 0;
    memcpy(hb + off_best, best, 8);
    GH_HIP(ctx, hipMemcpyAsync(d_models, hb + off_models, sizeof(m), hipMemcpyHostToDevice, ctx->stream));
    GH_HIP(ctx, hipMemcpyAsync(d_best, hb + off_best, 8, hipMemcpyHostToDevice, ctx->stream));
  } else {
    GH_LAUNCH(ctx, "ransac_solve", ransac_solve_kernel, dim3(kHyp / 64), dim3(64), 0, model, d_p, d_q, n, seed, nm,
              d_models, d_valid);
  }
  if (sampling == GH_SAMPLE_LMEDS) {
    // least median of squares (Rousseeuw; OpenCV's LMEDS): the hypothesis with the smallest median squared error wins
    // (lowest index on ties); inlier radius = max(threshold, 2.5 * 1.4826 * (1 + 5 / (n - s)) * sqrt(median))
    unsigned long long* d_med = (unsigned long long*)d_counts;
    GH_LAUNCH(ctx, "ransac_median", ransac_median_kernel, dim3(kHyp), dim3(256), 0, model, d_p, d_q, n, d_models, d_valid, d_med);
    const unsigned long long* h_med = (const unsigned long long*)(hb + off_counts);
    GH_HIP(ctx, hipMemcpyAsync(hb + off_counts, d_med, kHyp * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long bm = ~0ull;
    for (int h = 0; h < kHyp; ++h)
      if (h_med[h] < bm) {
        bm = h_med[h];
        best[0] = h;
      }
    if (hypotheses_used_out) *hypotheses_used_out = kHyp;
    if (best[0] < 0 || bm >= 0x7FF0000000000000ull) return GH_OK;  // no hypothesis explains half of the correspondences
    double med;
    memcpy(&med, &bm, 8);
    const double sigma = 2.5 * 1.4826 * (1.0 + 5.0 / (double)(n - s > 0 ? n - s : 1)) * sqrt(med);
    const double radius = sigma > threshold ? sigma : threshold;
    thr2 = radius * radius;
    memcpy(hb + off_best, best, 8);
    GH_HIP(ctx, hipMemcpyAsync(d_best, hb + off_best, 8, hipMemcpyHostToDevice, ctx->stream));
  } else if (sampling == GH_SAMPLE_RANSAC) {
    GH_LAUNCH(ctx, "ransac_score", ransac_score_kernel, dim3(kHyp), dim3(256), 0, model, d_p, d_q, n, thr2, d_models,
              d_valid, d_counts);
  }
  if (sampling != GH_SAMPLE_RANSAC) {
    // (the winner is known on the host already)
  } else if (confidence > 0.0 && confidence < 1.0) {
    // the prefix rule needs the counts on the host: 8 KB back, the choice (two ints) forth
    const int* h_counts = (const int*)(hb + off_counts);
    GH_HIP(ctx, hipMemcpyAsync(hb + off_counts, d_counts, kHyp * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int used = 0;
    best[0] = adaptive_prefix_best(h_counts, n, s, confidence, &best[1], &used);
    if (hypotheses_used_out) *hypotheses_used_out = used;
    // sent from the pinned mirror; the download below into the same words is ordered behind it by the stream
    memcpy(hb + off_best, best, 8);
    GH_HIP(ctx, hipMemcpyAsync(d_best, hb + off_best, 8, hipMemcpyHostToDevice, ctx->stream));
  } else {
    GH_LAUNCH(ctx, "ransac_best", ransac_best_kernel, dim3(1), dim3(256), 0, d_counts, d_best);
    if (hypotheses_used_out) *hypotheses_used_out = kHyp;
  }
  GH_LAUNCH(ctx, "ransac_mask", ransac_mask_kernel, dim3(gh_div_up(n > 12 ? n : 12, 256)), dim3(256), 0, model, d_p, d_q,
            n, thr2, d_models, d_best, d_mask, d_mout);
  // best | model | mask lie back to back: one download
  GH_HIP(ctx, hipMemcpyAsync(hb + off_best, d_best, (off_mask - off_best) + ((mask_out || sampling != GH_SAMPLE_RANSAC) ? (size_t)n : 0),
                             hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (sampling == GH_SAMPLE_RANSAC) memcpy(best, hb + off_best, 8);
  memcpy(model_out, hb + off_mout, 12 * 8);
  if (mask_out) memcpy(mask_out, hb + off_mask, (size_t)n);
  if (best[0] < 0) {
    for (int k = 0; k < 12; ++k) model_out[k] = 0.0;
    return GH_OK;
  }
  if (sampling != GH_SAMPLE_RANSAC) {  // the inlier count of the chosen model is the population of its mask
    best[1] = 0;
    const uint8_t* hm = hb + off_mask;
    for (int i = 0; i < n; ++i) best[1] += hm[i] ? 1 : 0;
  }
  *inliers_out = best[1];
  if (model == kModelE && !project_essential(model_out)) {  // the mask stays that of the scored 8-point estimate
    for (int k = 0; k < 12; ++k) model_out[k] = 0.0;
    *inliers_out = 0;
  }
  return GH_OK;
}

namespace {

// Midpoint triangulation of one correspondence per thread (GSLAM::Estimator::trianglate, Estimator.h:164-168): the point
// of the REFERENCE frame closest to both rays, X_cur = R X_ref + t.  ok = 0 when the rays are parallel or the point lies
// behind either camera.
__global__ __launch_bounds__(256) void triangulate_kernel(const double* __restrict__ pose, int pose_stride,
                                                          const double* __restrict__ dref, const double* __restrict__ dcur,
                                                          int n, double* __restrict__ out, uint8_t* __restrict__ ok) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double* T = pose + (size_t)i * pose_stride;  // [qx qy qz qw tx ty tz], ref -> cur
  const double qx = T[0], qy = T[1], qz = T[2], qw = T[3];
  const double R[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy),
                       2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx),
                       2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)};
  const double* d1 = dref + 3 * (size_t)i;
  const double* b = dcur + 3 * (size_t)i;
  double a[3];
  for (int r = 0; r < 3; ++r) a[r] = R[3 * r] * d1[0] + R[3 * r + 1] * d1[1] + R[3 * r + 2] * d1[2];
  const double aa = a[0] * a[0] + a[1] * a[1] + a[2] * a[2], bb = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
  const double ab = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
  const double at = a[0] * T[4] + a[1] * T[5] + a[2] * T[6], bt = b[0] * T[4] + b[1] * T[5] + b[2] * T[6];
  const double det = aa * bb - ab * ab;
  bool good = det > 1e-12 * aa * bb;
  double l1 = 0.0, l2 = 0.0;
  if (good) {
    l1 = (ab * bt - bb * at) / det;  // min |l1 a + t - l2 b|^2
    l2 = (aa * bt - ab * at) / det;
    good = l1 > 0.0 && l2 > 0.0;
  }
  double xr[3] = {0, 0, 0};
  if (good) {
    double mc[3];
    for (int r = 0; r < 3; ++r) mc[r] = ((l1 * a[r] + T[4 + r]) + l2 * b[r]) / 2.0 - T[4 + r];  // midpoint - t, cur frame
    for (int r = 0; r < 3; ++r) xr[r] = R[r] * mc[0] + R[3 + r] * mc[1] + R[6 + r] * mc[2];      // R^T (.)
  }
  for (int r = 0; r < 3; ++r) out[3 * (size_t)i + r] = xr[r];
  ok[i] = good ? 1 : 0;
}

}  // namespace

extern "C" gh_status gh_triangulate(gh_ctx* ctx, const double* ref2cur_pose, int pose_stride, const double* ref_dir,
                                    const double* cur_dir, int n, double* ref_points, uint8_t* ok) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, n >= 0 && (pose_stride == 0 || pose_stride == 7));
  if (n == 0) return GH_OK;
  GH_CHECK_ARG(ctx, ref2cur_pose && ref_dir && cur_dir && ref_points && ok);
  const size_t np_ = pose_stride == 0 ? 1 : (size_t)n;
  const size_t a = ((np_ * 56) + 255) & ~(size_t)255, b = (((size_t)n * 24) + 255) & ~(size_t)255;
  void *base = nullptr, *hbase = nullptr;
  const size_t total = a + 3 * b + (((size_t)n + 255) & ~(size_t)255);
  GH_TRY(gh_scratch(ctx, total, &base));
  GH_TRY(gh_pinned(ctx, total, &hbase));
  uint8_t *d = (uint8_t*)base, *hd = (uint8_t*)hbase;
  // [pose | ref_dir | cur_dir] up and [points | ok] down: one DMA each way through the pinned mirror of the layout
  memcpy(hd, ref2cur_pose, np_ * 56);
  memcpy(hd + a, ref_dir, (size_t)n * 24);
  memcpy(hd + a + b, cur_dir, (size_t)n * 24);
  GH_HIP(ctx, hipMemcpyAsync(d, hd, a + b + (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
  GH_LAUNCH(ctx, "triangulate", triangulate_kernel, dim3(gh_div_up(n, 256)), dim3(256), 0, (const double*)d, pose_stride,
            (const double*)(d + a), (const double*)(d + a + b), n, (double*)(d + a + 2 * b), d + a + 3 * b);
  GH_HIP(ctx, hipMemcpyAsync(hd + a + 2 * b, d + a + 2 * b, b + (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(ref_points, hd + a + 2 * b, (size_t)n * 24);
  memcpy(ok, hd + a + 3 * b, (size_t)n);
  return GH_OK;
}
