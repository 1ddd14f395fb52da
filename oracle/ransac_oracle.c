/*
 * ransac_oracle.c — CPU restatement of the RANSAC estimator.  TEST INFRASTRUCTURE ONLY.
 * PARITY UNPINNED: GSLAM ships only the Estimator INTERFACE (GSLAM/core/Estimator.h:92-169; the implementation
 * plugin is commented out of the build, CMakeLists.txt:45, and OpenCV is not installed), so there is no reference
 * arithmetic to pin.  Pinned: the interface shapes (3x3 H / F, 2x3 / 3x4 affine, uchar inlier mask).  The
 * specification below is this repo's; tests check it against ground-truth models on synthetic data.
 *
 *   hypotheses  2048, hypothesis h samples s distinct indices: state = sm64(seed ^ h * 0xD1B54A32D192ED03), then
 *               state = sm64(state), idx = state % n, redraw on duplicates            (sm64 = splitmix64 step)
 *   H  (s = 4)  8 x 8 system with h33 = 1, Gaussian elimination, partial pivoting (first maximum), pivot > 1e-12;
 *               error = |H x / w - x'|^2, undefined if |w| <= 1e-12
 *   A2 (s = 3)  3 x 3 system, two right-hand sides; error = |A [x y 1]^T - x'|^2
 *   A3 (s = 4)  4 x 4 system, three right-hand sides; error = |A [X Y Z 1]^T - X'|^2
 *   F  (s = 8)  Hartley normalisation of both sets over ALL points (mean, sqrt(2) / mean distance; sequential sums),
 *               8 x 9 nullspace by elimination with FULL pivoting (first maximum in row-major scan), free variable = 1,
 *               F = T2^T Fh T1; Sampson error (x'^T F x)^2 / (Fx_0^2 + Fx_1^2 + F^T x'_0^2 + F^T x'_1^2)
 *   E  (s = 8)  on normalised image coordinates: hypotheses and scoring exactly as F; the WINNER is projected onto the
 *               essential manifold: E = U diag(m, m, 0) V^T, m = (s1 + s2) / 2, through the cyclic-Jacobi
 *               eigen-decomposition of E^T E (12 sweeps, rotation t = sign(th) / (|th| + sqrt(th^2 + 1)))
 *   S  (s = 3)  SIM3 b ~ s R a + t by Horn's closed form (Horn 1987): centroids, S = sum a' b'^T, the 4 x 4 matrix N of
 *               the quaternion form, its dominant eigenvector by the same Jacobi routine (w >= 0), s = sqrt(|b'|^2 /
 *               |a'|^2), t = cb - s R ca; model [qx qy qz qw tx ty tz s]; error |s R a + t - b|^2
 *   P  (s = 3)  plane through three points, unit normal, model [n d]; error (n . x + d)^2
 *   PnP (s = 6) direct linear transform: 12 x 12 nullspace by full-pivot elimination, scaled to |r3| = 1 with the first
 *               sample point in front of the camera, rows made orthonormal by Gram-Schmidt (reflections rejected);
 *               model [R | t]; error = squared reprojection distance on the z = 1 plane, depth > 1e-12
 *   winner      most correspondences with error <= threshold^2; lowest hypothesis index on ties
 * All arithmetic is IEEE double without FMA contraction, in the order written here (the GPU kernels keep the same order).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define R_HYP 2048
#define R_TINY 1e-12

static uint64_t sm64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static int ge(double a[8][12], int n, int nrhs) {
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double best = fabs(a[k][k]);
    for (int r = k + 1; r < n; ++r)
      if (fabs(a[r][k]) > best) {
        best = fabs(a[r][k]);
        piv = r;
      }
    if (!(best > R_TINY)) return 0;
    if (piv != k)
      for (int c = 0; c < n + nrhs; ++c) {
        double t = a[k][c];
        a[k][c] = a[piv][c];
        a[piv][c] = t;
      }
    double inv = 1.0 / a[k][k];
    for (int r = k + 1; r < n; ++r) {
      double f = a[r][k] * inv;
      for (int c = k; c < n + nrhs; ++c) a[r][c] = a[r][c] - f * a[k][c];
    }
  }
  for (int j = 0; j < nrhs; ++j)
    for (int r = n - 1; r >= 0; --r) {
      double s = a[r][n + j];
      for (int c = r + 1; c < n; ++c) s = s - a[r][c] * a[c][n + j];
      a[r][n + j] = s / a[r][r];
    }
  return 1;
}

typedef struct {
  double m1x, m1y, s1, m2x, m2y, s2;
} norm_t;


static void jacobi3(double a[3][3], double v[3][3]);
static void jacobi4(double a[4][4], double v[4][4]);
#define JACOBI_BODY(N)                                                                     \
  for (int i = 0; i < N; ++i)                                                              \
    for (int j = 0; j < N; ++j) v[i][j] = i == j ? 1.0 : 0.0;                               \
  for (int sweep = 0; sweep < 12; ++sweep)                                                  \
    for (int p = 0; p < N - 1; ++p)                                                         \
      for (int q = p + 1; q < N; ++q) {                                                     \
        double apq = a[p][q];                                                               \
        if (!(fabs(apq) > 1e-300)) continue;                                                \
        double theta = (a[q][q] - a[p][p]) / (2.0 * apq);                                   \
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));   \
        double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;                                     \
        for (int k = 0; k < N; ++k) {                                                       \
          double akp = a[k][p], akq = a[k][q];                                              \
          a[k][p] = c * akp - sn * akq;                                                     \
          a[k][q] = sn * akp + c * akq;                                                     \
        }                                                                                   \
        for (int k = 0; k < N; ++k) {                                                       \
          double apk = a[p][k], aqk = a[q][k];                                              \
          a[p][k] = c * apk - sn * aqk;                                                     \
          a[q][k] = sn * apk + c * aqk;                                                     \
        }                                                                                   \
        for (int k = 0; k < N; ++k) {                                                       \
          double vkp = v[k][p], vkq = v[k][q];                                              \
          v[k][p] = c * vkp - sn * vkq;                                                     \
          v[k][q] = sn * vkp + c * vkq;                                                     \
        }                                                                                   \
      }
static void jacobi3(double a[3][3], double v[3][3]) { JACOBI_BODY(3) }
static void jacobi4(double a[4][4], double v[4][4]) { JACOBI_BODY(4) }
static void jacobi9(double a[9][9], double v[9][9]) { JACOBI_BODY(9) }
static void jacobi12(double a[12][12], double v[12][12]) { JACOBI_BODY(12) }

static void quat_R(double qx, double qy, double qz, double qw, double* R) {
  R[0] = 1 - 2 * (qy * qy + qz * qz); R[1] = 2 * (qx * qy - qw * qz); R[2] = 2 * (qx * qz + qw * qy);
  R[3] = 2 * (qx * qy + qw * qz); R[4] = 1 - 2 * (qx * qx + qz * qz); R[5] = 2 * (qy * qz - qw * qx);
  R[6] = 2 * (qx * qz - qw * qy); R[7] = 2 * (qy * qz + qw * qx); R[8] = 1 - 2 * (qx * qx + qy * qy);
}

/* m pairs: the three of a sample through idx, or all n in index order with idx = NULL (NOSAMPLE) */
static int solve_sim3_m(const double* p, const double* q, const int* idx, double* out, int m);
static int solve_sim3(const double* p, const double* q, const int* idx, double* out) { return solve_sim3_m(p, q, idx, out, 3); }
static int solve_sim3_m(const double* p, const double* q, const int* idxp, double* out, int m) {
  double ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0};
  for (int j = 0; j < m; ++j)
    for (int e = 0; e < 3; ++e) {
      ca[e] = ca[e] + p[3 * (idxp ? idxp[j] : j) + e];
      cb[e] = cb[e] + q[3 * (idxp ? idxp[j] : j) + e];
    }
  for (int e = 0; e < 3; ++e) {
    ca[e] = ca[e] / (double)m;
    cb[e] = cb[e] / (double)m;
  }
  double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, na = 0.0, nb = 0.0;
  for (int j = 0; j < m; ++j) {
    double a[3], b[3];
    for (int e = 0; e < 3; ++e) {
      a[e] = p[3 * (idxp ? idxp[j] : j) + e] - ca[e];
      b[e] = q[3 * (idxp ? idxp[j] : j) + e] - cb[e];
      na = na + a[e] * a[e];
      nb = nb + b[e] * b[e];
    }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) S[r][c] = S[r][c] + a[r] * b[c];
  }
  if (!(na > R_TINY) || !(nb > R_TINY)) return 0;
  double N[4][4] = {{S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0]},
                    {0, S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2]},
                    {0, 0, -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1]},
                    {0, 0, 0, -S[0][0] - S[1][1] + S[2][2]}};
  for (int r = 1; r < 4; ++r)
    for (int c = 0; c < r; ++c) N[r][c] = N[c][r];
  double V[4][4];
  jacobi4(N, V);
  int best = 0;
  for (int k = 1; k < 4; ++k)
    if (N[k][k] > N[best][best]) best = k;
  double qw = V[0][best], qx = V[1][best], qy = V[2][best], qz = V[3][best];
  double qn = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
  if (!(qn > R_TINY)) return 0;
  if (qw < 0) { qw = -qw; qx = -qx; qy = -qy; qz = -qz; }
  qw = qw / qn; qx = qx / qn; qy = qy / qn; qz = qz / qn;
  double sc = sqrt(nb / na), R[9];
  quat_R(qx, qy, qz, qw, R);
  out[0] = qx; out[1] = qy; out[2] = qz; out[3] = qw;
  for (int r = 0; r < 3; ++r) out[4 + r] = cb[r] - sc * (R[3 * r] * ca[0] + R[3 * r + 1] * ca[1] + R[3 * r + 2] * ca[2]);
  out[7] = sc;
  return 1;
}

static int solve_plane(const double* p, const int* idx, double* out) {
  const double *p0 = p + 3 * idx[0], *p1 = p + 3 * idx[1], *p2 = p + 3 * idx[2];
  double u[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, v[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
  double n[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
  double len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  if (!(len > R_TINY)) return 0;
  for (int e = 0; e < 3; ++e) n[e] = n[e] / len;
  out[0] = n[0]; out[1] = n[1]; out[2] = n[2];
  out[3] = -(n[0] * p0[0] + n[1] * p0[1] + n[2] * p0[2]);
  return 1;
}

static int pnp_from_projection(double* P, const double* X0, double* out);
static int solve_pnp_dlt(const double* p, const double* q, const int* idx, double* out) {
  double a[12][12];
  for (int j = 0; j < 6; ++j) {
    double X = p[3 * idx[j]], Y = p[3 * idx[j] + 1], Z = p[3 * idx[j] + 2], u = q[2 * idx[j]], v = q[2 * idx[j] + 1];
    double r0[12] = {X, Y, Z, 1, 0, 0, 0, 0, -u * X, -u * Y, -u * Z, -u};
    double r1[12] = {0, 0, 0, 0, X, Y, Z, 1, -v * X, -v * Y, -v * Z, -v};
    memcpy(a[2 * j], r0, sizeof(r0));
    memcpy(a[2 * j + 1], r1, sizeof(r1));
  }
  int perm[12];
  for (int c = 0; c < 12; ++c) perm[c] = c;
  for (int k = 0; k < 11; ++k) {
    int pr = k, pc = k;
    double best = -1.0;
    for (int r = k; r < 12; ++r)
      for (int c = k; c < 12; ++c)
        if (fabs(a[r][c]) > best) {
          best = fabs(a[r][c]);
          pr = r;
          pc = c;
        }
    if (!(best > R_TINY)) return 0;
    if (pr != k)
      for (int c = 0; c < 12; ++c) {
        double t = a[k][c];
        a[k][c] = a[pr][c];
        a[pr][c] = t;
      }
    if (pc != k) {
      for (int r = 0; r < 12; ++r) {
        double t = a[r][k];
        a[r][k] = a[r][pc];
        a[r][pc] = t;
      }
      int t = perm[k];
      perm[k] = perm[pc];
      perm[pc] = t;
    }
    double inv = 1.0 / a[k][k];
    for (int r = k + 1; r < 12; ++r) {
      double f = a[r][k] * inv;
      for (int c = k; c < 12; ++c) a[r][c] = a[r][c] - f * a[k][c];
    }
  }
  double z[12], P[12];
  z[11] = 1.0;
  for (int r = 10; r >= 0; --r) {
    double s = 0.0;
    for (int c = r + 1; c < 12; ++c) s = s + a[r][c] * z[c];
    z[r] = -s / a[r][r];
  }
  for (int c = 0; c < 12; ++c) P[perm[c]] = z[c];
  return pnp_from_projection(P, p + 3 * idx[0], out);
}

static int pnp_from_projection(double* P, const double* X0, double* out) {
  double n3 = sqrt(P[8] * P[8] + P[9] * P[9] + P[10] * P[10]);
  if (!(n3 > R_TINY)) return 0;
  double lam = 1.0 / n3;
  if ((P[8] * X0[0] + P[9] * X0[1] + P[10] * X0[2] + P[11]) * lam < 0) lam = -lam;
  for (int c = 0; c < 12; ++c) P[c] = P[c] * lam;
  double r1[3] = {P[0], P[1], P[2]}, r2[3] = {P[4], P[5], P[6]};
  double n1 = sqrt(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]);
  if (!(n1 > R_TINY)) return 0;
  for (int e = 0; e < 3; ++e) r1[e] = r1[e] / n1;
  double d12 = r2[0] * r1[0] + r2[1] * r1[1] + r2[2] * r1[2];
  for (int e = 0; e < 3; ++e) r2[e] = r2[e] - d12 * r1[e];
  double n2 = sqrt(r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2]);
  if (!(n2 > R_TINY)) return 0;
  for (int e = 0; e < 3; ++e) r2[e] = r2[e] / n2;
  double r3[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
  if (!(r3[0] * P[8] + r3[1] * P[9] + r3[2] * P[10] > 0)) return 0;
  for (int e = 0; e < 3; ++e) {
    out[e] = r1[e];
    out[3 + e] = r2[e];
    out[6 + e] = r3[e];
  }
  out[9] = P[3] / n1;
  out[10] = P[7] / n2;
  out[11] = P[11];
  return 1;
}

static int project_essential(double* E) {
  double B[3][3], V[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) acc = acc + E[3 * k + r] * E[3 * k + c];
      B[r][c] = acc;
    }
  jacobi3(B, V);
  int o[3] = {0, 1, 2};
  for (int a = 0; a < 2; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (B[o[b]][o[b]] > B[o[a]][o[a]]) {
        int t = o[a];
        o[a] = o[b];
        o[b] = t;
      }
  double l1 = B[o[0]][o[0]], l2 = B[o[1]][o[1]];
  if (!(l2 > 1e-300)) return 0;
  double s1 = sqrt(l1), s2 = sqrt(l2), sm = (s1 + s2) / 2.0, u[2][3];
  for (int a = 0; a < 2; ++a) {
    double sv = a == 0 ? s1 : s2;
    for (int r = 0; r < 3; ++r) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) acc = acc + E[3 * r + k] * V[k][o[a]];
      u[a][r] = acc / sv;
    }
  }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) E[3 * r + c] = sm * (u[0][r] * V[c][o[0]] + u[1][r] * V[c][o[1]]);
  return 1;
}

static int solve_model(int model, const double* p, const double* q, const int* idx, const norm_t* nm, double* out) {
  if (model == 5) return solve_sim3(p, q, idx, out);
  if (model == 6) return solve_plane(p, idx, out);
  if (model == 7) return solve_pnp_dlt(p, q, idx, out);
  double a[8][12];
  memset(a, 0, sizeof(a));
  if (model == 0) {
    for (int j = 0; j < 4; ++j) {
      double x = p[2 * idx[j]], y = p[2 * idx[j] + 1], u = q[2 * idx[j]], v = q[2 * idx[j] + 1];
      double r0[9] = {x, y, 1, 0, 0, 0, -u * x, -u * y, u}, r1[9] = {0, 0, 0, x, y, 1, -v * x, -v * y, v};
      memcpy(a[2 * j], r0, sizeof(r0));
      memcpy(a[2 * j + 1], r1, sizeof(r1));
    }
    if (!ge(a, 8, 1)) return 0;
    for (int i = 0; i < 8; ++i) out[i] = a[i][8];
    out[8] = 1.0;
    return 1;
  }
  if (model == 1) {
    for (int j = 0; j < 3; ++j) {
      a[j][0] = p[2 * idx[j]]; a[j][1] = p[2 * idx[j] + 1]; a[j][2] = 1;
      a[j][3] = q[2 * idx[j]]; a[j][4] = q[2 * idx[j] + 1];
    }
    if (!ge(a, 3, 2)) return 0;
    for (int i = 0; i < 3; ++i) {
      out[i] = a[i][3];
      out[3 + i] = a[i][4];
    }
    return 1;
  }
  if (model == 3) {
    for (int j = 0; j < 4; ++j) {
      a[j][0] = p[3 * idx[j]]; a[j][1] = p[3 * idx[j] + 1]; a[j][2] = p[3 * idx[j] + 2]; a[j][3] = 1;
      a[j][4] = q[3 * idx[j]]; a[j][5] = q[3 * idx[j] + 1]; a[j][6] = q[3 * idx[j] + 2];
    }
    if (!ge(a, 4, 3)) return 0;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) out[4 * r + c] = a[c][4 + r];
    return 1;
  }
  for (int j = 0; j < 8; ++j) {
    double x = (p[2 * idx[j]] - nm->m1x) * nm->s1, y = (p[2 * idx[j] + 1] - nm->m1y) * nm->s1;
    double u = (q[2 * idx[j]] - nm->m2x) * nm->s2, v = (q[2 * idx[j] + 1] - nm->m2y) * nm->s2;
    double r[9] = {u * x, u * y, u, v * x, v * y, v, x, y, 1};
    memcpy(a[j], r, sizeof(r));
  }
  int perm[9];
  for (int c = 0; c < 9; ++c) perm[c] = c;
  for (int k = 0; k < 8; ++k) {
    int pr = k, pc = k;
    double best = -1.0;
    for (int r = k; r < 8; ++r)
      for (int c = k; c < 9; ++c)
        if (fabs(a[r][c]) > best) {
          best = fabs(a[r][c]);
          pr = r;
          pc = c;
        }
    if (!(best > R_TINY)) return 0;
    if (pr != k)
      for (int c = 0; c < 9; ++c) {
        double t = a[k][c];
        a[k][c] = a[pr][c];
        a[pr][c] = t;
      }
    if (pc != k) {
      for (int r = 0; r < 8; ++r) {
        double t = a[r][k];
        a[r][k] = a[r][pc];
        a[r][pc] = t;
      }
      int t = perm[k];
      perm[k] = perm[pc];
      perm[pc] = t;
    }
    double inv = 1.0 / a[k][k];
    for (int r = k + 1; r < 8; ++r) {
      double f = a[r][k] * inv;
      for (int c = k; c < 9; ++c) a[r][c] = a[r][c] - f * a[k][c];
    }
  }
  double z[9], fh[9], tmp[9];
  z[8] = 1.0;
  for (int r = 7; r >= 0; --r) {
    double s = 0.0;
    for (int c = r + 1; c < 9; ++c) s = s + a[r][c] * z[c];
    z[r] = -s / a[r][r];
  }
  for (int c = 0; c < 9; ++c) fh[perm[c]] = z[c];
  double T1[9] = {nm->s1, 0, -nm->s1 * nm->m1x, 0, nm->s1, -nm->s1 * nm->m1y, 0, 0, 1};
  double T2[9] = {nm->s2, 0, -nm->s2 * nm->m2x, 0, nm->s2, -nm->s2 * nm->m2y, 0, 0, 1};
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
  return 1;
}

static int model_err(int model, const double* m, const double* p, const double* q, int i, double* err) {
  if (model == 0) {
    double x = p[2 * i], y = p[2 * i + 1];
    double w = m[6] * x + m[7] * y + m[8];
    if (!(fabs(w) > R_TINY)) return 0;
    double px = (m[0] * x + m[1] * y + m[2]) / w, py = (m[3] * x + m[4] * y + m[5]) / w;
    double dx = px - q[2 * i], dy = py - q[2 * i + 1];
    *err = dx * dx + dy * dy;
    return 1;
  }
  if (model == 1) {
    double x = p[2 * i], y = p[2 * i + 1];
    double dx = (m[0] * x + m[1] * y + m[2]) - q[2 * i], dy = (m[3] * x + m[4] * y + m[5]) - q[2 * i + 1];
    *err = dx * dx + dy * dy;
    return 1;
  }
  if (model == 3) {
    double X = p[3 * i], Y = p[3 * i + 1], Z = p[3 * i + 2], e = 0.0;
    for (int r = 0; r < 3; ++r) {
      double d = (m[4 * r] * X + m[4 * r + 1] * Y + m[4 * r + 2] * Z + m[4 * r + 3]) - q[3 * i + r];
      e = e + d * d;
    }
    *err = e;
    return 1;
  }
  if (model == 5) {
    double R[9], X = p[3 * i], Y = p[3 * i + 1], Z = p[3 * i + 2], e = 0.0;
    quat_R(m[0], m[1], m[2], m[3], R);
    for (int r = 0; r < 3; ++r) {
      double d = (m[7] * (R[3 * r] * X + R[3 * r + 1] * Y + R[3 * r + 2] * Z) + m[4 + r]) - q[3 * i + r];
      e = e + d * d;
    }
    *err = e;
    return 1;
  }
  if (model == 6) {
    double d = m[0] * p[3 * i] + m[1] * p[3 * i + 1] + m[2] * p[3 * i + 2] + m[3];
    *err = d * d;
    return 1;
  }
  if (model == 7) {
    double X = p[3 * i], Y = p[3 * i + 1], Z = p[3 * i + 2];
    double zc = m[6] * X + m[7] * Y + m[8] * Z + m[11];
    if (!(zc > R_TINY)) return 0;
    double dx = (m[0] * X + m[1] * Y + m[2] * Z + m[9]) / zc - q[2 * i];
    double dy = (m[3] * X + m[4] * Y + m[5] * Z + m[10]) / zc - q[2 * i + 1];
    *err = dx * dx + dy * dy;
    return 1;
  }
  double x = p[2 * i], y = p[2 * i + 1], u = q[2 * i], v = q[2 * i + 1];
  double fx0 = m[0] * x + m[1] * y + m[2], fx1 = m[3] * x + m[4] * y + m[5], fx2 = m[6] * x + m[7] * y + m[8];
  double ft0 = m[0] * u + m[3] * v + m[6], ft1 = m[1] * u + m[4] * v + m[7];
  double num = u * fx0 + v * fx1 + fx2;
  double den = fx0 * fx0 + fx1 * fx1 + ft0 * ft0 + ft1 * ft1;
  if (!(den > 1e-300)) return 0;
  *err = (num * num) / den;
  return 1;
}

int oracle_ransac_conf(int model, const double* p, const double* q, int n, double threshold, double confidence,
                       uint64_t seed, double* model_out, uint8_t* mask, int* used_out);

/* returns the inlier count of the winning hypothesis (0 = no model) */
int oracle_ransac(int model, const double* p, const double* q, int n, double threshold, uint64_t seed, double* model_out,
                  uint8_t* mask) {
  return oracle_ransac_conf(model, p, q, n, threshold, 1.0, seed, model_out, mask, 0);
}

/* GSLAM::Estimator's confidence argument (Estimator.h:100-169): plain sequential RANSAC with the textbook adaptive
 * stopping rule -- after a strict improvement of the best inlier count c, N = ceil(log(1 - conf) / log(1 - (c/n)^s));
 * stop when h + 1 >= N.  confidence outside (0, 1): all R_HYP hypotheses. */
int oracle_ransac_conf(int model, const double* p, const double* q, int n, double threshold, double confidence,
                       uint64_t seed, double* model_out, uint8_t* mask, int* used_out) {
  static const int S_OF[8] = {4, 3, 8, 4, 8, 3, 3, 6}, M_OF[8] = {9, 6, 9, 12, 9, 8, 4, 12};
  const int s = S_OF[model], ms = M_OF[model];
  memset(model_out, 0, 12 * sizeof(double));
  if (mask) memset(mask, 0, (size_t)(n > 0 ? n : 0));
  if (n < s) return 0;
  norm_t nm = {0, 0, 1, 0, 0, 1};
  if (model == 2 || model == 4) {
    double ax = 0, ay = 0, bx = 0, by = 0;
    for (int i = 0; i < n; ++i) {
      ax += p[2 * i]; ay += p[2 * i + 1];
      bx += q[2 * i]; by += q[2 * i + 1];
    }
    nm.m1x = ax / n; nm.m1y = ay / n; nm.m2x = bx / n; nm.m2y = by / n;
    double d1 = 0, d2 = 0;
    for (int i = 0; i < n; ++i) {
      double x = p[2 * i] - nm.m1x, y = p[2 * i + 1] - nm.m1y, u = q[2 * i] - nm.m2x, v = q[2 * i + 1] - nm.m2y;
      d1 += sqrt(x * x + y * y);
      d2 += sqrt(u * u + v * v);
    }
    d1 /= n; d2 /= n;
    nm.s1 = d1 > 0 ? 1.4142135623730951 / d1 : 1.0;
    nm.s2 = d2 > 0 ? 1.4142135623730951 / d2 : 1.0;
  }
  const double thr2 = threshold * threshold;
  int best_h = -1, best_c = -1, limit = R_HYP, h = 0;
  const int adaptive = confidence > 0.0 && confidence < 1.0;
  double best_m[12];
  if (used_out) *used_out = 0;
  for (; h < limit; ++h) {
    int idx[8];
    uint64_t st = sm64(seed ^ ((uint64_t)h * 0xD1B54A32D192ED03ull));
    for (int j = 0; j < s; ++j)
      for (;;) {
        st = sm64(st);
        int c = (int)(st % (uint64_t)n), dup = 0;
        for (int t = 0; t < j; ++t) dup |= idx[t] == c;
        if (!dup) {
          idx[j] = c;
          break;
        }
      }
    double m[12];
    memset(m, 0, sizeof(m));
    if (!solve_model(model, p, q, idx, &nm, m)) continue;
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
      double e;
      if (model_err(model, m, p, q, i, &e) && e <= thr2) ++cnt;
    }
    if (cnt > best_c) {
      best_c = cnt;
      best_h = h;
      memcpy(best_m, m, sizeof(m));
      if (adaptive && best_c > 0) {
        double pg = pow((double)best_c / (double)n, (double)s);
        int need = R_HYP;
        if (pg >= 1.0) need = h + 1;
        else if (pg > 0.0) {
          double v = ceil(log(1.0 - confidence) / log(1.0 - pg));
          need = v < 1.0 ? 1 : (v > (double)R_HYP ? R_HYP : (int)v);
        }
        if (need < limit) limit = need < h + 1 ? h + 1 : need;
      }
    }
  }
  if (used_out) *used_out = h;
  if (best_h < 0) return 0;
  for (int k = 0; k < ms; ++k) model_out[k] = best_m[k];
  if (model == 4 && !project_essential(model_out)) { /* the mask stays that of the scored 8-point estimate */
    memset(model_out, 0, 12 * sizeof(double));
    if (mask) memset(mask, 0, (size_t)n);
    return 0;
  }
  if (mask)
    for (int i = 0; i < n; ++i) {
      double e;
      mask[i] = (uint8_t)((model_err(model, best_m, p, q, i, &e) && e <= thr2) ? 1 : 0);
    }
  return best_c;
}

/* ---- sampling modes of GSLAM::EstimatorMethod (Estimator.h:86-89) beside RANSAC: LMEDS and NOSAMPLE -- the restatement
 * gh_ransac_estimate_ex is checked against.  sampling: 0 RANSAC (= oracle_ransac_conf), 1 LMEDS, 2 NOSAMPLE. */
static int cmp_double(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

static void hartley(int model, const double* p, const double* q, int n, norm_t* nm) {
  nm->m1x = nm->m1y = nm->m2x = nm->m2y = 0;
  nm->s1 = nm->s2 = 1;
  if (model != 2 && model != 4) return;
  double ax = 0, ay = 0, bx = 0, by = 0;
  for (int i = 0; i < n; ++i) {
    ax += p[2 * i]; ay += p[2 * i + 1];
    bx += q[2 * i]; by += q[2 * i + 1];
  }
  nm->m1x = ax / n; nm->m1y = ay / n; nm->m2x = bx / n; nm->m2y = by / n;
  double d1 = 0, d2 = 0;
  for (int i = 0; i < n; ++i) {
    double x = p[2 * i] - nm->m1x, y = p[2 * i + 1] - nm->m1y, u = q[2 * i] - nm->m2x, v = q[2 * i + 1] - nm->m2y;
    d1 += sqrt(x * x + y * y);
    d2 += sqrt(u * u + v * v);
  }
  d1 /= n; d2 /= n;
  nm->s1 = d1 > 0 ? 1.4142135623730951 / d1 : 1.0;
  nm->s2 = d2 > 0 ? 1.4142135623730951 / d2 : 1.0;
}

/* NOSAMPLE: the algebraic least-squares model of all correspondences, sums in index order */
static int fit_all(int model, const double* p, const double* q, int n, const norm_t* nm, double* out) {
  memset(out, 0, 12 * sizeof(double));
  if (model == 0 || model == 1 || model == 3) {
    int nu = model == 0 ? 8 : (model == 1 ? 3 : 4), nr = model == 0 ? 1 : (model == 1 ? 2 : 3);
    double N[8][12];
    memset(N, 0, sizeof(N));
    for (int i = 0; i < n; ++i) {
      double rows[2][9];
      int nrows = 1;
      if (model == 0) {
        double x = p[2 * i], y = p[2 * i + 1], u = q[2 * i], v = q[2 * i + 1];
        double r0[9] = {x, y, 1, 0, 0, 0, -u * x, -u * y, u}, r1[9] = {0, 0, 0, x, y, 1, -v * x, -v * y, v};
        memcpy(rows[0], r0, sizeof(r0));
        memcpy(rows[1], r1, sizeof(r1));
        nrows = 2;
      } else if (model == 1) {
        double r[9] = {p[2 * i], p[2 * i + 1], 1, q[2 * i], q[2 * i + 1], 0, 0, 0, 0};
        memcpy(rows[0], r, sizeof(r));
      } else {
        double r[9] = {p[3 * i], p[3 * i + 1], p[3 * i + 2], 1, q[3 * i], q[3 * i + 1], q[3 * i + 2], 0, 0};
        memcpy(rows[0], r, sizeof(r));
      }
      for (int k = 0; k < nrows; ++k)
        for (int a = 0; a < nu; ++a)
          for (int b = 0; b < nu + nr; ++b) N[a][b] = N[a][b] + rows[k][a] * rows[k][b];
    }
    if (!ge(N, nu, nr)) return 0;
    if (model == 0) {
      for (int k = 0; k < 8; ++k) out[k] = N[k][8];
      out[8] = 1.0;
    } else if (model == 1) {
      for (int k = 0; k < 3; ++k) {
        out[k] = N[k][3];
        out[3 + k] = N[k][4];
      }
    } else {
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) out[4 * r + c] = N[c][4 + r];
    }
    return 1;
  }
  if (model == 5) return solve_sim3_m(p, q, 0, out, n);
  if (model == 6) {
    double c[3] = {0, 0, 0}, C[3][3], V[3][3];
    memset(C, 0, sizeof(C));
    for (int i = 0; i < n; ++i)
      for (int e = 0; e < 3; ++e) c[e] = c[e] + p[3 * i + e];
    for (int e = 0; e < 3; ++e) c[e] = c[e] / (double)n;
    for (int i = 0; i < n; ++i) {
      double d[3] = {p[3 * i] - c[0], p[3 * i + 1] - c[1], p[3 * i + 2] - c[2]};
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) C[a][b] = C[a][b] + d[a] * d[b];
    }
    jacobi3(C, V);
    int best = 0;
    for (int k = 1; k < 3; ++k)
      if (C[k][k] < C[best][best]) best = k;
    double nv[3] = {V[0][best], V[1][best], V[2][best]};
    double len = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
    if (!(len > R_TINY)) return 0;
    for (int e = 0; e < 3; ++e) out[e] = nv[e] / len;
    out[3] = -(out[0] * c[0] + out[1] * c[1] + out[2] * c[2]);
    return 1;
  }
  if (model == 7) {
    double A[12][12], V[12][12];
    memset(A, 0, sizeof(A));
    for (int i = 0; i < n; ++i) {
      double X = p[3 * i], Y = p[3 * i + 1], Z = p[3 * i + 2], u = q[2 * i], v = q[2 * i + 1];
      double r0[12] = {X, Y, Z, 1, 0, 0, 0, 0, -u * X, -u * Y, -u * Z, -u}, r1[12] = {0, 0, 0, 0, X, Y, Z, 1, -v * X, -v * Y, -v * Z, -v};
      for (int a = 0; a < 12; ++a)
        for (int b = 0; b < 12; ++b) A[a][b] = A[a][b] + (r0[a] * r0[b] + r1[a] * r1[b]);
    }
    jacobi12(A, V);
    int best = 0;
    for (int k = 1; k < 12; ++k)
      if (A[k][k] < A[best][best]) best = k;
    double P[12];
    for (int k = 0; k < 12; ++k) P[k] = V[k][best];
    return pnp_from_projection(P, p, out);
  }
  double A[9][9], V[9][9];
  memset(A, 0, sizeof(A));
  for (int i = 0; i < n; ++i) {
    double x = (p[2 * i] - nm->m1x) * nm->s1, y = (p[2 * i + 1] - nm->m1y) * nm->s1;
    double u = (q[2 * i] - nm->m2x) * nm->s2, v = (q[2 * i + 1] - nm->m2y) * nm->s2;
    double r[9] = {u * x, u * y, u, v * x, v * y, v, x, y, 1};
    for (int a = 0; a < 9; ++a)
      for (int b = 0; b < 9; ++b) A[a][b] = A[a][b] + r[a] * r[b];
  }
  jacobi9(A, V);
  int best = 0;
  for (int k = 1; k < 9; ++k)
    if (A[k][k] < A[best][best]) best = k;
  double fh[9];
  for (int k = 0; k < 9; ++k) fh[k] = V[k][best];
  double T1[9] = {nm->s1, 0, -nm->s1 * nm->m1x, 0, nm->s1, -nm->s1 * nm->m1y, 0, 0, 1};
  double T2[9] = {nm->s2, 0, -nm->s2 * nm->m2x, 0, nm->s2, -nm->s2 * nm->m2y, 0, 0, 1};
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
  return 1;
}

/* returns the inlier count (0 = no model) */
int oracle_estimate_ex(int model, const double* p, const double* q, int n, double threshold, double confidence,
                       uint64_t seed, int sampling, double* model_out, uint8_t* mask, int* used_out) {
  static const int S_OF[8] = {4, 3, 8, 4, 8, 3, 3, 6}, M_OF[8] = {9, 6, 9, 12, 9, 8, 4, 12};
  if (sampling == 0) return oracle_ransac_conf(model, p, q, n, threshold, confidence, seed, model_out, mask, used_out);
  const int s = S_OF[model], ms = M_OF[model];
  memset(model_out, 0, 12 * sizeof(double));
  if (mask) memset(mask, 0, (size_t)(n > 0 ? n : 0));
  if (used_out) *used_out = 0;
  if (n < s) return 0;
  norm_t nm;
  hartley(model, p, q, n, &nm);
  double best_m[12], thr2 = threshold * threshold;
  memset(best_m, 0, sizeof(best_m));
  if (sampling == 2) {
    if (!fit_all(model, p, q, n, &nm, best_m)) return 0;
    if (used_out) *used_out = 1;
  } else {
    double best_med = INFINITY;
    int best_h = -1;
    double* errs = (double*)malloc(sizeof(double) * (size_t)n);
    for (int h = 0; h < R_HYP; ++h) {
      int idx[8];
      uint64_t st = sm64(seed ^ ((uint64_t)h * 0xD1B54A32D192ED03ull));
      for (int j = 0; j < s; ++j)
        for (;;) {
          st = sm64(st);
          int c = (int)(st % (uint64_t)n), dup = 0;
          for (int t = 0; t < j; ++t) dup |= idx[t] == c;
          if (!dup) {
            idx[j] = c;
            break;
          }
        }
      double m[12];
      memset(m, 0, sizeof(m));
      if (!solve_model(model, p, q, idx, &nm, m)) continue;
      for (int i = 0; i < n; ++i) {
        double e;
        errs[i] = (model_err(model, m, p, q, i, &e) && e == e) ? e : INFINITY;
      }
      qsort(errs, (size_t)n, sizeof(double), cmp_double);
      double med = errs[n / 2];
      if (med < best_med) { /* strict: the lowest index wins ties */
        best_med = med;
        best_h = h;
        memcpy(best_m, m, sizeof(m));
      }
    }
    free(errs);
    if (used_out) *used_out = R_HYP;
    if (best_h < 0 || !(best_med < INFINITY)) return 0;
    double sigma = 2.5 * 1.4826 * (1.0 + 5.0 / (double)(n - s > 0 ? n - s : 1)) * sqrt(best_med);
    double radius = sigma > threshold ? sigma : threshold;
    thr2 = radius * radius;
  }
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    double e;
    int in = (model_err(model, best_m, p, q, i, &e) && e <= thr2) ? 1 : 0;
    if (mask) mask[i] = (uint8_t)in;
    cnt += in;
  }
  for (int k = 0; k < ms; ++k) model_out[k] = best_m[k];
  if (model == 4 && !project_essential(model_out)) {
    memset(model_out, 0, 12 * sizeof(double));
    if (mask) memset(mask, 0, (size_t)n);
    return 0;
  }
  return cnt;
}

/* Midpoint triangulation (GSLAM::Estimator::trianglate, Estimator.h:164-168): the point of the reference frame closest
 * to the rays d_ref and d_cur, X_cur = R X_ref + t, pose = [qx qy qz qw tx ty tz].  Returns 0 for parallel rays or a
 * point behind either camera. */
int oracle_triangulate(const double* T, const double* d1, const double* b, double* out) {
  double R[9], a[3];
  quat_R(T[0], T[1], T[2], T[3], R);
  for (int r = 0; r < 3; ++r) a[r] = R[3 * r] * d1[0] + R[3 * r + 1] * d1[1] + R[3 * r + 2] * d1[2];
  double aa = a[0] * a[0] + a[1] * a[1] + a[2] * a[2], bb = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
  double ab = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
  double at = a[0] * T[4] + a[1] * T[5] + a[2] * T[6], bt = b[0] * T[4] + b[1] * T[5] + b[2] * T[6];
  double det = aa * bb - ab * ab;
  out[0] = out[1] = out[2] = 0.0;
  if (!(det > 1e-12 * aa * bb)) return 0;
  double l1 = (ab * bt - bb * at) / det, l2 = (aa * bt - ab * at) / det;
  if (!(l1 > 0.0 && l2 > 0.0)) return 0;
  double mc[3];
  for (int r = 0; r < 3; ++r) mc[r] = ((l1 * a[r] + T[4 + r]) + l2 * b[r]) / 2.0 - T[4 + r];
  for (int r = 0; r < 3; ++r) out[r] = R[r] * mc[0] + R[3 + r] * mc[1] + R[6 + r] * mc[2];
  return 1;
}
