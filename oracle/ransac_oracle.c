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
 *   winner      most correspondences with error <= threshold^2; lowest hypothesis index on ties
 * All arithmetic is IEEE double without FMA contraction, in the order written here (the GPU kernels keep the same order).
 */
#include <math.h>
#include <stdint.h>
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

static int solve_model(int model, const double* p, const double* q, const int* idx, const norm_t* nm, double* out) {
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
  double x = p[2 * i], y = p[2 * i + 1], u = q[2 * i], v = q[2 * i + 1];
  double fx0 = m[0] * x + m[1] * y + m[2], fx1 = m[3] * x + m[4] * y + m[5], fx2 = m[6] * x + m[7] * y + m[8];
  double ft0 = m[0] * u + m[3] * v + m[6], ft1 = m[1] * u + m[4] * v + m[7];
  double num = u * fx0 + v * fx1 + fx2;
  double den = fx0 * fx0 + fx1 * fx1 + ft0 * ft0 + ft1 * ft1;
  if (!(den > 1e-300)) return 0;
  *err = (num * num) / den;
  return 1;
}

/* returns the inlier count of the winning hypothesis (0 = no model) */
int oracle_ransac(int model, const double* p, const double* q, int n, double threshold, uint64_t seed, double* model_out,
                  uint8_t* mask) {
  const int s = model == 0 ? 4 : (model == 1 ? 3 : (model == 2 ? 8 : 4));
  const int ms = model == 0 ? 9 : (model == 1 ? 6 : (model == 2 ? 9 : 12));
  memset(model_out, 0, 12 * sizeof(double));
  if (mask) memset(mask, 0, (size_t)(n > 0 ? n : 0));
  if (n < s) return 0;
  norm_t nm = {0, 0, 1, 0, 0, 1};
  if (model == 2) {
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
  int best_h = -1, best_c = -1;
  double best_m[12];
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
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
      double e;
      if (model_err(model, m, p, q, i, &e) && e <= thr2) ++cnt;
    }
    if (cnt > best_c) {
      best_c = cnt;
      best_h = h;
      memcpy(best_m, m, sizeof(m));
    }
  }
  if (best_h < 0) return 0;
  for (int k = 0; k < ms; ++k) model_out[k] = best_m[k];
  if (mask)
    for (int i = 0; i < n; ++i) {
      double e;
      mask[i] = (uint8_t)((model_err(model, best_m, p, q, i, &e) && e <= thr2) ? 1 : 0);
    }
  return best_c;
}
