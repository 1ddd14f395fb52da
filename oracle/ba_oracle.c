/*
 * ba_oracle.c — CPU restatement of the Levenberg-Marquardt bundle-adjustment inner loop behind
 * GSLAM::Optimizer::optimize(BundleGraph&).  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED for the solver itself: the reference's implementation is the Ceres plugin
 * (GSLAM/plugins/optimizer_ceres), which is absent from the tree (CMakeLists.txt:44 commented out)
 * and Ceres/Eigen are not installed, no version pinned (.travis.yml:161-184 installs "whatever apt
 * has").  Pinned pieces, followed line by line:
 *   GSLAM/core/Optimizer.h:102-182  problem container: keyframes = T_wc (camera->world) + dof bits,
 *                                   mappoints = (xyz, notFixed), BundleEdge = {pointId, frameId,
 *                                   measurement on the z=1 plane, information 2x2 or NULL}
 *   GSLAM/core/Optimizer.h:70-84    UPDATE_KF_* bits -> masked Jacobian columns
 *   GSLAM/core/Optimizer.h:174-182  projectErrorHuberThreshold (0.01), maxIterations (500)
 *   GSLAM/core/SE3.h:100-103        inverse;  :257-287 exp([v, w]) (translation first), with the
 *                                   translation coefficients GUARDED near theta = 0 (the reference
 *                                   returns NaN for w == 0)
 *   GSLAM/core/SO3.h:489-509        quaternion product / rotation of a point (30-flop form)
 *   tests/golden/se3_reference.npz  holds exp/log/mul/inverse/apply outputs of the reference's own
 *                                   SE3 (generated through oracle/_ref) that pin the pose algebra here.
 * The trust-region policy restates Ceres' published Levenberg-Marquardt strategy
 * (ceres/levenberg_marquardt_strategy.cc, trust_region_minimizer.cc; Agarwal et al., "Bundle
 * Adjustment in the Large", 2010 for the Schur elimination):
 *   damping D = clamp(diag(J^T J), 1e-6, 1e32) / radius;  initial radius 1e4;
 *   rho = (cost - cost_new) / model_cost_change;  accept if rho > 1e-3 then
 *   radius /= max(1/3, 1 - (2 rho - 1)^3), decrease_factor = 2; else radius /= decrease_factor,
 *   decrease_factor *= 2;  stop on |dcost| <= 1e-6 cost, max|g| <= 1e-10, or max iterations.
 *   Huber(delta) enters as the IRLS weight rho'(s) (Ceres' corrector with rho'' <= 0).
 *   Observations with camera-frame depth <= 1e-9 contribute nothing (GSLAM/core/Camera.h:213-245 rejects z <= 0); a
 *   candidate step that moves a previously valid observation behind its camera is rejected outright.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BA_MAX_TRACE 512

typedef struct {
  double huber_delta;
  int32_t max_iterations;
  double initial_radius, function_tolerance, gradient_tolerance, min_relative_decrease;
  int32_t verbose, deterministic;
} oracle_ba_options;

typedef struct {
  int32_t iterations, accepted, termination;
  double initial_cost, final_cost, solve_ms_total, total_ms;
  int32_t trace_len;
  double trace_cost[BA_MAX_TRACE], trace_radius[BA_MAX_TRACE];
  uint8_t trace_accepted[BA_MAX_TRACE];
} oracle_ba_summary;

/* ---------------------------------------------------------------- pose algebra (pose = qx qy qz qw tx ty tz) */
static void quat_rotate(const double* q, const double* p, double* o) { /* SO3.h:497-509 */
  double uvx = q[1] * p[2] - q[2] * p[1], uvy = q[2] * p[0] - q[0] * p[2], uvz = q[0] * p[1] - q[1] * p[0];
  uvx += uvx; uvy += uvy; uvz += uvz;
  o[0] = p[0] + q[3] * uvx + (q[1] * uvz - q[2] * uvy);
  o[1] = p[1] + q[3] * uvy + (q[2] * uvx - q[0] * uvz);
  o[2] = p[2] + q[3] * uvz + (q[0] * uvy - q[1] * uvx);
}

static void quat_mul(const double* a, const double* b, double* o) { /* SO3.h:489-495 */
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}

/* SE3::exp (SE3.h:257-287) with guarded translation coefficients. xi = [v(3), w(3)] */
void oracle_se3_exp(const double* xi, double* pose) {
  const double* v = xi;
  const double* w = xi + 3;
  double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double th = sqrt(th2);
  double imag, real, A, B;
  if (th < 1e-5) {
    double th4 = th2 * th2;
    imag = 0.5 - th2 / 48.0 + th4 / 3840.0;
    real = 1.0 - th2 / 8.0 + th4 / 384.0;
    A = 0.5 - th2 / 24.0 + th4 / 720.0;          /* (1 - cos th) / th^2 */
    B = 1.0 / 6.0 - th2 / 120.0 + th4 / 5040.0;  /* (th - sin th) / th^3 */
  } else {
    imag = sin(0.5 * th) / th;
    real = cos(0.5 * th);
    A = (1.0 - cos(th)) / th2;
    B = (th - sin(th)) / (th2 * th);
  }
  pose[0] = imag * w[0];
  pose[1] = imag * w[1];
  pose[2] = imag * w[2];
  pose[3] = real;
  double c1[3] = {w[1] * v[2] - w[2] * v[1], w[2] * v[0] - w[0] * v[2], w[0] * v[1] - w[1] * v[0]};
  double c2[3] = {w[1] * c1[2] - w[2] * c1[1], w[2] * c1[0] - w[0] * c1[2], w[0] * c1[1] - w[1] * c1[0]};
  for (int i = 0; i < 3; ++i) pose[4 + i] = v[i] + A * c1[i] + B * c2[i];
}

/* T <- T * exp(xi)  (SE3.h:120-123) and renormalise the quaternion */
void oracle_se3_retract(const double* pose, const double* xi, double* out) {
  double e[7], q[4], t[3];
  oracle_se3_exp(xi, e);
  quat_mul(pose, e, q);
  quat_rotate(pose, e + 4, t);
  double n = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) out[i] = q[i] * n;
  for (int i = 0; i < 3; ++i) out[4 + i] = pose[4 + i] + t[i];
}

/* X_c = T_wc^-1 X_w = R^T (X_w - t)  (SE3.h:100-103) */
static void world_to_cam(const double* pose, const double* X, double* Xc) {
  double qc[4] = {-pose[0], -pose[1], -pose[2], pose[3]};
  double d[3] = {X[0] - pose[4], X[1] - pose[5], X[2] - pose[6]};
  quat_rotate(qc, d, Xc);
}

static void rot_matrix_T(const double* q, double* Rt) { /* R^T, row-major; R as SO3.h:360-374 */
  double x = q[0], y = q[1], z = q[2], w = q[3];
  double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rt[i * 3 + j] = R[j * 3 + i];
}

#define BA_MIN_DEPTH 1e-9

/* One observation: weighted residual pieces.  Returns 0 if the point is not in front of the camera.
 * r[2], w (IRLS weight), Jc[2][6] (masked by dof), Jp[2][3] (zero if point fixed), s = r^T L r. */
static int obs_linearize(const double* pose, int dof, const double* X, int pfree, const double* m, const double* info,
                         double huber, double* r, double* wgt, double* Jc, double* Jp, double* s_out) {
  double Xc[3];
  world_to_cam(pose, X, Xc);
  if (!(Xc[2] > BA_MIN_DEPTH)) return 0;
  double iz = 1.0 / Xc[2];
  double u = Xc[0] * iz, v = Xc[1] * iz;
  r[0] = u - m[0];
  r[1] = v - m[1];
  double L00 = 1, L01 = 0, L10 = 0, L11 = 1;
  if (info) { L00 = info[0]; L01 = info[1]; L10 = info[2]; L11 = info[3]; }
  double s = r[0] * (L00 * r[0] + L01 * r[1]) + r[1] * (L10 * r[0] + L11 * r[1]);
  double w = 1.0;
  if (huber > 0 && s > huber * huber) w = huber / sqrt(s);
  *wgt = w;
  *s_out = s;
  if (!Jc) return 1;
  /* d pi / d Xc */
  double P[6] = {iz, 0, -u * iz, 0, iz, -v * iz};
  /* d Xc = -dv + [Xc]x dw */
  double D[18] = {-1, 0, 0, 0, -Xc[2], Xc[1],
                  0, -1, 0, Xc[2], 0, -Xc[0],
                  0, 0, -1, -Xc[1], Xc[0], 0};
  for (int a = 0; a < 2; ++a)
    for (int k = 0; k < 6; ++k) {
      double acc = 0;
      for (int j = 0; j < 3; ++j) acc += P[a * 3 + j] * D[j * 6 + k];
      Jc[a * 6 + k] = ((dof >> k) & 1) ? acc : 0.0;
    }
  double Rt[9];
  rot_matrix_T(pose, Rt);
  for (int a = 0; a < 2; ++a)
    for (int k = 0; k < 3; ++k) {
      double acc = 0;
      for (int j = 0; j < 3; ++j) acc += P[a * 3 + j] * Rt[j * 3 + k];
      Jp[a * 3 + k] = pfree ? acc : 0.0;
    }
  return 1;
}

static double rho_huber(double s, double huber) {
  if (huber > 0 && s > huber * huber) return 2.0 * huber * sqrt(s) - huber * huber;
  return s;
}

typedef struct {
  int nc, np, no;
  const int32_t* dof;
  const uint8_t* pfree;
  const int32_t *ocam, *opt;
  const double *oxy, *oinfo;
  double huber;
} ba_ctx;

static double total_cost(const ba_ctx* c, const double* poses, const double* pts) {
  double cost = 0;
  for (int k = 0; k < c->no; ++k) {
    double r[2], w, s;
    if (!obs_linearize(poses + 7 * c->ocam[k], 0, pts + 3 * c->opt[k], 0, c->oxy + 2 * k,
                       c->oinfo ? c->oinfo + 4 * k : NULL, c->huber, r, &w, NULL, NULL, &s))
      continue;
    cost += rho_huber(s, c->huber);
  }
  return 0.5 * cost;
}

/* Dense lower Cholesky, in place, column-major n x n with leading dimension n; returns 0 or failing column+1.
 * Blocked right-looking with OpenMP on the trailing update (only the timed baseline needs the speed). */
int oracle_potrf(double* A, int n, int threads) {
  const int NB = 64;
  for (int k0 = 0; k0 < n; k0 += NB) {
    int kb = n - k0 < NB ? n - k0 : NB;
    for (int j = k0; j < k0 + kb; ++j) { /* unblocked on the diagonal block + panel column scaling */
      double d = A[(size_t)j * n + j];
      for (int t = k0; t < j; ++t) d -= A[(size_t)t * n + j] * A[(size_t)t * n + j];
      if (!(d > 0)) return j + 1;
      d = sqrt(d);
      A[(size_t)j * n + j] = d;
      double inv = 1.0 / d;
#pragma omp parallel for num_threads(threads) schedule(static) if (n - j > 512)
      for (int i = j + 1; i < n; ++i) {
        double v = A[(size_t)j * n + i];
        for (int t = k0; t < j; ++t) v -= A[(size_t)t * n + i] * A[(size_t)t * n + j];
        A[(size_t)j * n + i] = v * inv;
      }
    }
    int r0 = k0 + kb;
    /* trailing update A22 -= L21 L21^T (lower part) */
#pragma omp parallel for num_threads(threads) schedule(dynamic, 8)
    for (int j = r0; j < n; ++j) {
      double* cj = A + (size_t)j * n;
      for (int t = k0; t < k0 + kb; ++t) {
        const double* ct = A + (size_t)t * n;
        double ljt = ct[j];
        for (int i = j; i < n; ++i) cj[i] -= ct[i] * ljt;
      }
    }
  }
  return 0;
}

void oracle_potrs(const double* L, int n, double* b) {
  for (int j = 0; j < n; ++j) {
    b[j] /= L[(size_t)j * n + j];
    double bj = b[j];
    for (int i = j + 1; i < n; ++i) b[i] -= L[(size_t)j * n + i] * bj;
  }
  for (int j = n - 1; j >= 0; --j) {
    double v = b[j];
    for (int i = j + 1; i < n; ++i) v -= L[(size_t)j * n + i] * b[i];
    b[j] = v / L[(size_t)j * n + j];
  }
}

static int inv3_sym(const double* H, double* Hi) { /* H row-major 3x3 SPD */
  double a = H[0], b = H[1], c = H[2], d = H[4], e = H[5], f = H[8];
  double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  double det = a * c00 + b * c01 + c * c02;
  if (!(det > 0)) return 0;
  double id = 1.0 / det;
  Hi[0] = c00 * id; Hi[1] = c01 * id; Hi[2] = c02 * id;
  Hi[3] = Hi[1]; Hi[4] = (a * f - c * c) * id; Hi[5] = (b * c - a * e) * id;
  Hi[6] = Hi[2]; Hi[7] = Hi[5]; Hi[8] = (a * d - b * b) * id;
  return 1;
}

static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Full LM solve.  poses: nc x 7 [qx qy qz qw tx ty tz] in/out; pts: np x 3 in/out. */
int oracle_ba_solve(int nc, int np, int no, double* poses, const int32_t* dof, double* pts, const uint8_t* pfree,
                    const int32_t* ocam, const int32_t* opt, const double* oxy, const double* oinfo,
                    const oracle_ba_options* opt_in, oracle_ba_summary* sum, int threads) {
  ba_ctx c = {nc, np, no, dof, pfree, ocam, opt, oxy, oinfo, opt_in->huber_delta};
  const int n = 6 * nc;
  double* Hcc = (double*)calloc((size_t)nc * 36, 8);
  double* gc = (double*)calloc((size_t)n, 8);
  double* Hpp = (double*)calloc((size_t)np * 9, 8);
  double* gp = (double*)calloc((size_t)np * 3, 8);
  double* Hpi = (double*)calloc((size_t)np * 9, 8);
  double* S = (double*)malloc((size_t)n * n * 8);
  double* dc = (double*)calloc((size_t)n, 8);
  double* dp = (double*)calloc((size_t)np * 3, 8);
  double* poses_new = (double*)malloc((size_t)nc * 7 * 8);
  double* pts_new = (double*)malloc((size_t)np * 3 * 8);
  /* observation lists per point */
  int* pstart = (int*)calloc((size_t)np + 1, sizeof(int));
  int* plist = (int*)malloc((size_t)(no > 0 ? no : 1) * sizeof(int));
  for (int k = 0; k < no; ++k) pstart[opt[k] + 1]++;
  for (int p = 0; p < np; ++p) pstart[p + 1] += pstart[p];
  {
    int* fill = (int*)calloc((size_t)np, sizeof(int));
    for (int k = 0; k < no; ++k) plist[pstart[opt[k]] + fill[opt[k]]++] = k;
    free(fill);
  }
  memset(sum, 0, sizeof(*sum));
  double radius = opt_in->initial_radius, decrease = 2.0;
  double cost = total_cost(&c, poses, pts);
  sum->initial_cost = cost;
  int need_lin = 1, term = 0, it = 0;
  for (it = 0; it < opt_in->max_iterations; ++it) {
    if (need_lin) {
      memset(Hcc, 0, (size_t)nc * 36 * 8);
      memset(gc, 0, (size_t)n * 8);
      memset(Hpp, 0, (size_t)np * 9 * 8);
      memset(gp, 0, (size_t)np * 3 * 8);
      for (int k = 0; k < no; ++k) {
        int ci = ocam[k], pi = opt[k];
        double r[2], w, s, Jc[12], Jp[6];
        const double* info = oinfo ? oinfo + 4 * k : NULL;
        if (!obs_linearize(poses + 7 * ci, dof[ci], pts + 3 * pi, pfree ? pfree[pi] : 1, oxy + 2 * k, info,
                           c.huber, r, &w, Jc, Jp, &s))
          continue;
        double L[4] = {w, 0, 0, w};
        if (info) { L[0] = w * info[0]; L[1] = w * info[1]; L[2] = w * info[2]; L[3] = w * info[3]; }
        double Lr[2] = {L[0] * r[0] + L[1] * r[1], L[2] * r[0] + L[3] * r[1]};
        double LJc[12], LJp[6];
        for (int j = 0; j < 6; ++j) {
          LJc[j] = L[0] * Jc[j] + L[1] * Jc[6 + j];
          LJc[6 + j] = L[2] * Jc[j] + L[3] * Jc[6 + j];
        }
        for (int j = 0; j < 3; ++j) {
          LJp[j] = L[0] * Jp[j] + L[1] * Jp[3 + j];
          LJp[3 + j] = L[2] * Jp[j] + L[3] * Jp[3 + j];
        }
        for (int a = 0; a < 6; ++a) {
          gc[6 * ci + a] += Jc[a] * Lr[0] + Jc[6 + a] * Lr[1];
          for (int b = 0; b < 6; ++b) Hcc[36 * ci + 6 * a + b] += Jc[a] * LJc[b] + Jc[6 + a] * LJc[6 + b];
        }
        for (int a = 0; a < 3; ++a) {
          gp[3 * pi + a] += Jp[a] * Lr[0] + Jp[3 + a] * Lr[1];
          for (int b = 0; b < 3; ++b) Hpp[9 * pi + 3 * a + b] += Jp[a] * LJp[b] + Jp[3 + a] * LJp[3 + b];
        }
      }
      double gmax = 0;
      for (int i = 0; i < n; ++i) gmax = fmax(gmax, fabs(gc[i]));
      for (int i = 0; i < 3 * np; ++i) gmax = fmax(gmax, fabs(gp[i]));
      if (gmax <= opt_in->gradient_tolerance) { term = 2; break; }
      need_lin = 0;
    }
    /* damped blocks */
    int ok = 1;
    for (int p = 0; p < np; ++p) {
      double H[9];
      memcpy(H, Hpp + 9 * p, 72);
      for (int a = 0; a < 3; ++a) H[4 * a] += clampd(Hpp[9 * p + 4 * a], 1e-6, 1e32) / radius;
      if (!inv3_sym(H, Hpi + 9 * p)) ok = 0;
    }
    memset(S, 0, (size_t)n * n * 8);
    for (int ci = 0; ci < nc; ++ci)
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) {
          double v = Hcc[36 * ci + 6 * a + b];
          if (a == b) v += clampd(v, 1e-6, 1e32) / radius;
          S[(size_t)(6 * ci + b) * n + 6 * ci + a] = v; /* column-major */
        }
    for (int i = 0; i < n; ++i) dc[i] = -gc[i];
    /* Schur complement: per point, all pairs of its observations */
    for (int p = 0; p < np && ok; ++p) {
      int n_o = pstart[p + 1] - pstart[p];
      if (n_o == 0) continue;
      double* Wb = (double*)malloc((size_t)n_o * 18 * 8); /* W_i = Jc^T L Jp, 6x3 */
      double* WH = (double*)malloc((size_t)n_o * 18 * 8); /* W_i Hpp^-1 */
      int* cams = (int*)malloc((size_t)n_o * sizeof(int));
      int m = 0;
      for (int q = 0; q < n_o; ++q) {
        int k = plist[pstart[p] + q];
        int ci = ocam[k];
        double r[2], w, s, Jc[12], Jp[6];
        const double* info = oinfo ? oinfo + 4 * k : NULL;
        if (!obs_linearize(poses + 7 * ci, dof[ci], pts + 3 * p, pfree ? pfree[p] : 1, oxy + 2 * k, info, c.huber, r,
                           &w, Jc, Jp, &s))
          continue;
        double L[4] = {w, 0, 0, w};
        if (info) { L[0] = w * info[0]; L[1] = w * info[1]; L[2] = w * info[2]; L[3] = w * info[3]; }
        double LJp[6];
        for (int j = 0; j < 3; ++j) {
          LJp[j] = L[0] * Jp[j] + L[1] * Jp[3 + j];
          LJp[3 + j] = L[2] * Jp[j] + L[3] * Jp[3 + j];
        }
        for (int a = 0; a < 6; ++a)
          for (int b = 0; b < 3; ++b) Wb[18 * m + 3 * a + b] = Jc[a] * LJp[b] + Jc[6 + a] * LJp[3 + b];
        for (int a = 0; a < 6; ++a)
          for (int b = 0; b < 3; ++b) {
            double acc = 0;
            for (int t = 0; t < 3; ++t) acc += Wb[18 * m + 3 * a + t] * Hpi[9 * p + 3 * t + b];
            WH[18 * m + 3 * a + b] = acc;
          }
        cams[m++] = ci;
      }
      for (int i = 0; i < m; ++i) {
        for (int a = 0; a < 6; ++a) {
          double acc = 0;
          for (int t = 0; t < 3; ++t) acc += WH[18 * i + 3 * a + t] * gp[3 * p + t];
          dc[6 * cams[i] + a] += acc;
        }
        for (int j = 0; j < m; ++j)
          for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b) {
              double acc = 0;
              for (int t = 0; t < 3; ++t) acc += WH[18 * i + 3 * a + t] * Wb[18 * j + 3 * b + t];
              S[(size_t)(6 * cams[j] + b) * n + 6 * cams[i] + a] -= acc;
            }
      }
      free(Wb); free(WH); free(cams);
    }
    if (ok && oracle_potrf(S, n, threads) != 0) ok = 0;
    double new_cost = cost, model = 0, rho = -1;
    if (ok) {
      oracle_potrs(S, n, dc);
      /* back-substitution and model cost change, one pass over the observations of each point */
      for (int p = 0; p < np; ++p) {
        double rhs[3] = {-gp[3 * p], -gp[3 * p + 1], -gp[3 * p + 2]};
        for (int q = pstart[p]; q < pstart[p + 1]; ++q) {
          int k = plist[q], ci = ocam[k];
          double r[2], w, s, Jc[12], Jp[6];
          const double* info = oinfo ? oinfo + 4 * k : NULL;
          if (!obs_linearize(poses + 7 * ci, dof[ci], pts + 3 * p, pfree ? pfree[p] : 1, oxy + 2 * k, info, c.huber,
                             r, &w, Jc, Jp, &s))
            continue;
          double L[4] = {w, 0, 0, w};
          if (info) { L[0] = w * info[0]; L[1] = w * info[1]; L[2] = w * info[2]; L[3] = w * info[3]; }
          double Jd[2] = {0, 0};
          for (int a = 0; a < 6; ++a) { Jd[0] += Jc[a] * dc[6 * ci + a]; Jd[1] += Jc[6 + a] * dc[6 * ci + a]; }
          double LJd[2] = {L[0] * Jd[0] + L[1] * Jd[1], L[2] * Jd[0] + L[3] * Jd[1]};
          for (int b = 0; b < 3; ++b) rhs[b] -= Jp[b] * LJd[0] + Jp[3 + b] * LJd[1]; /* W^T dc */
        }
        for (int a = 0; a < 3; ++a)
          dp[3 * p + a] = Hpi[9 * p + 3 * a] * rhs[0] + Hpi[9 * p + 3 * a + 1] * rhs[1] + Hpi[9 * p + 3 * a + 2] * rhs[2];
      }
      for (int k = 0; k < no; ++k) {
        int ci = ocam[k], pi = opt[k];
        double r[2], w, s, Jc[12], Jp[6];
        const double* info = oinfo ? oinfo + 4 * k : NULL;
        if (!obs_linearize(poses + 7 * ci, dof[ci], pts + 3 * pi, pfree ? pfree[pi] : 1, oxy + 2 * k, info, c.huber,
                           r, &w, Jc, Jp, &s))
          continue;
        double L[4] = {w, 0, 0, w};
        if (info) { L[0] = w * info[0]; L[1] = w * info[1]; L[2] = w * info[2]; L[3] = w * info[3]; }
        double Jd[2] = {0, 0};
        for (int a = 0; a < 6; ++a) { Jd[0] += Jc[a] * dc[6 * ci + a]; Jd[1] += Jc[6 + a] * dc[6 * ci + a]; }
        for (int a = 0; a < 3; ++a) { Jd[0] += Jp[a] * dp[3 * pi + a]; Jd[1] += Jp[3 + a] * dp[3 * pi + a]; }
        double LJd[2] = {L[0] * Jd[0] + L[1] * Jd[1], L[2] * Jd[0] + L[3] * Jd[1]};
        double Lr[2] = {L[0] * r[0] + L[1] * r[1], L[2] * r[0] + L[3] * r[1]};
        model -= Jd[0] * Lr[0] + Jd[1] * Lr[1] + 0.5 * (Jd[0] * LJd[0] + Jd[1] * LJd[1]);
      }
      for (int ci = 0; ci < nc; ++ci) {
        if ((dof[ci] & 63) == 0) memcpy(poses_new + 7 * ci, poses + 7 * ci, 56); /* fixed: bitwise untouched */
        else oracle_se3_retract(poses + 7 * ci, dc + 6 * ci, poses_new + 7 * ci);
      }
      for (int i = 0; i < 3 * np; ++i) pts_new[i] = pts[i] + dp[i];
      new_cost = total_cost(&c, poses_new, pts_new);
      /* An observation that was in front of its camera at the linearisation point and is not at the candidate would
       * silently leave the sum and LOWER the cost: such a candidate is rejected (infinite cost) instead. */
      for (int k = 0; k < no; ++k) {
        double r[2], w, s;
        const double* info = oinfo ? oinfo + 4 * k : NULL;
        if (obs_linearize(poses + 7 * ocam[k], 0, pts + 3 * opt[k], 0, oxy + 2 * k, info, c.huber, r, &w, NULL, NULL, &s) &&
            !obs_linearize(poses_new + 7 * ocam[k], 0, pts_new + 3 * opt[k], 0, oxy + 2 * k, info, c.huber, r, &w, NULL,
                           NULL, &s)) {
          new_cost = INFINITY;
          break;
        }
      }
      rho = model > 0 ? (cost - new_cost) / model : -1;
    }
    int acc = ok && rho > opt_in->min_relative_decrease;
    if (sum->trace_len < BA_MAX_TRACE) {
      sum->trace_cost[sum->trace_len] = new_cost;
      sum->trace_radius[sum->trace_len] = radius;
      sum->trace_accepted[sum->trace_len] = (uint8_t)acc;
      sum->trace_len++;
    }
    if (acc) {
      double dcost = cost - new_cost;
      memcpy(poses, poses_new, (size_t)nc * 7 * 8);
      memcpy(pts, pts_new, (size_t)np * 3 * 8);
      double t = 2.0 * rho - 1.0;
      radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
      if (radius > 1e16) radius = 1e16;
      decrease = 2.0;
      sum->accepted++;
      need_lin = 1;
      double prev = cost;
      cost = new_cost;
      if (fabs(dcost) <= opt_in->function_tolerance * prev) { term = 1; ++it; break; }
    } else {
      radius = radius / decrease;
      decrease *= 2.0;
      if (radius < 1e-32) { term = 3; ++it; break; }
    }
  }
  sum->iterations = it;
  sum->termination = term;
  sum->final_cost = cost;
  free(Hcc); free(gc); free(Hpp); free(gp); free(Hpi); free(S); free(dc); free(dp);
  free(poses_new); free(pts_new); free(pstart); free(plist);
  return term == 3 ? 1 : 0;
}

double oracle_ba_cost(int nc, int np, int no, const double* poses, const double* pts, const int32_t* ocam,
                      const int32_t* opt, const double* oxy, const double* oinfo, double huber) {
  ba_ctx c = {nc, np, no, NULL, NULL, ocam, opt, oxy, oinfo, huber};
  return total_cost(&c, poses, pts);
}

/* Exposes the per-observation linearisation to the tests (finite-difference Jacobian check, scipy cross-check). */
int oracle_ba_obs_linearize(const double* pose, int dof, const double* X, int pfree, const double* m,
                            const double* info, double huber, double* r, double* wgt, double* Jc, double* Jp,
                            double* s_out) {
  return obs_linearize(pose, dof, X, pfree, m, info, huber, r, wgt, Jc, Jp, s_out);
}

/* Pose-only (motion-only) BA behind GSLAM::Optimizer::optimizePnP (GSLAM/core/Optimizer.h:202-207): the same LM on a
 * 1-camera graph whose points are all fixed; information_out (may be NULL) = J^T W J at the solution, row-major 6x6,
 * W = Huber IRLS weight, columns masked by dof.  pose in/out [qx qy qz qw tx ty tz]. */
int oracle_ba_pnp(const double* points_xyz, const double* obs_xy, int n, double* pose, int dof,
                  const oracle_ba_options* opt, double* information_out, oracle_ba_summary* sum) {
  int nn = n > 0 ? n : 1;
  int32_t* ocam = (int32_t*)calloc((size_t)nn, sizeof(int32_t));
  int32_t* opt_idx = (int32_t*)malloc((size_t)nn * sizeof(int32_t));
  uint8_t* pfree = (uint8_t*)calloc((size_t)nn, 1);
  double* pts = (double*)malloc((size_t)nn * 3 * 8);
  memcpy(pts, points_xyz, (size_t)n * 3 * 8);
  for (int i = 0; i < n; ++i) opt_idx[i] = i;
  int32_t d = dof;
  int rc = oracle_ba_solve(1, n, n, pose, &d, pts, pfree, ocam, opt_idx, obs_xy, NULL, opt, sum, 1);
  if (information_out) {
    memset(information_out, 0, 36 * 8);
    for (int k = 0; k < n; ++k) {
      double r[2], w, s, Jc[12], Jp[6];
      if (!obs_linearize(pose, dof, points_xyz + 3 * k, 0, obs_xy + 2 * k, NULL, opt->huber_delta, r, &w, Jc, Jp, &s))
        continue;
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) information_out[6 * a + b] += w * (Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b]);
    }
  }
  free(ocam); free(opt_idx); free(pfree); free(pts);
  return rc;
}


/* ---------------------------------------------------------------------------------------------------------------------
 * Optimizer::magin (GSLAM/core/Optimizer.h:230-232: "Convert bundle graph to pose graph").  The reference declares it and
 * nothing else -- no implementation, no caller, no test -- so this is the specification (parity unpinned):
 *   for every pair of cameras first < second that share at least min_shared points (a point counts once per pair of its
 *   observations in the two cameras), one SE3 edge {first, second, T_first^-1 T_second, Lambda};
 *   Lambda (6 x 6, [v, w] order of SE3::exp, row-major) = the information of the second camera's pose, in the perturbation
 *   T <- T exp(delta), relative to the first camera held fixed, from the two-view problem over the shared points at the
 *   current estimate:   Lambda = sum_p  A - B V^-1 B^T   with, for the observation of p in `second`,
 *       A = Jc^T L Jc,  B = Jc^T L Jp,  and  V = Jp^T L Jp summed over BOTH observations of p   (L = Huber weight x information);
 *   a fixed point (or a singular V) contributes A alone; a point behind either camera contributes nothing.
 * Edges sorted by (first, second).  Returns the number of edges (all of them, also beyond max_edges: then only the first
 * max_edges are written). */
typedef struct { int64_t key; int32_t k1, k2, seq; } marg_entry;
static int marg_cmp(const void* a, const void* b) {
  const marg_entry* x = (const marg_entry*)a;
  const marg_entry* y = (const marg_entry*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0);
}
int oracle_ba_marginalize(int nc, int np, int no, const double* poses, const double* pts, const uint8_t* pfree,
                          const int32_t* ocam, const int32_t* opt, const double* oxy, const double* oinfo, double huber,
                          int min_shared, int max_edges, int32_t* edge_first, int32_t* edge_second, int32_t* edge_shared,
                          double* edge_info) {
  int32_t* pstart = (int32_t*)calloc((size_t)np + 1, sizeof(int32_t));
  int32_t* plist = (int32_t*)malloc((size_t)(no > 0 ? no : 1) * sizeof(int32_t));
  int32_t* cur = (int32_t*)malloc((size_t)(np > 0 ? np : 1) * sizeof(int32_t));
  for (int k = 0; k < no; ++k) ++pstart[opt[k] + 1];
  for (int p = 0; p < np; ++p) pstart[p + 1] += pstart[p];
  for (int p = 0; p < np; ++p) cur[p] = pstart[p];
  for (int k = 0; k < no; ++k) plist[cur[opt[k]]++] = k;
  size_t cap = 0;
  for (int p = 0; p < np; ++p) {
    const size_t d = (size_t)(pstart[p + 1] - pstart[p]);
    cap += d * (d - (d > 0)) / 2;
  }
  marg_entry* ent = (marg_entry*)malloc((cap > 0 ? cap : 1) * sizeof(marg_entry));
  size_t ne = 0;
  for (int p = 0; p < np; ++p)
    for (int a = pstart[p]; a < pstart[p + 1]; ++a)
      for (int b = a + 1; b < pstart[p + 1]; ++b) {
        const int ka = plist[a], kb = plist[b], ca = ocam[ka], cb = ocam[kb];
        if (ca == cb) continue;
        ent[ne].key = (int64_t)(ca < cb ? ca : cb) * nc + (ca < cb ? cb : ca);
        ent[ne].k1 = ca < cb ? ka : kb;
        ent[ne].k2 = ca < cb ? kb : ka;
        ent[ne].seq = (int32_t)ne;
        ++ne;
      }
  qsort(ent, ne, sizeof(marg_entry), marg_cmp);
  int n_edges = 0;
  for (size_t b0 = 0; b0 < ne;) {
    size_t b1 = b0;
    while (b1 < ne && ent[b1].key == ent[b0].key) ++b1;
    if ((int64_t)(b1 - b0) >= min_shared) {
      const int ci = (int)(ent[b0].key / nc), cj = (int)(ent[b0].key % nc);
      if (n_edges < max_edges) {
        double Lam[36];
        for (int t = 0; t < 36; ++t) Lam[t] = 0;
        for (size_t e = b0; e < b1; ++e) {
          const int k1 = ent[e].k1, k2 = ent[e].k2, p = opt[k2];
          const int pf = pfree ? pfree[p] : 1;
          double r1[2], w1, Jc1[12], Jp1[6], s1, r2[2], w2, Jc2[12], Jp2[6], s2;
          if (!obs_linearize(poses + 7 * ci, 63, pts + 3 * p, pf, oxy + 2 * k1, oinfo ? oinfo + 4 * k1 : NULL, huber, r1, &w1, Jc1, Jp1, &s1)) continue;
          if (!obs_linearize(poses + 7 * cj, 63, pts + 3 * p, pf, oxy + 2 * k2, oinfo ? oinfo + 4 * k2 : NULL, huber, r2, &w2, Jc2, Jp2, &s2)) continue;
          double L1[4] = {w1, 0, 0, w1}, L2[4] = {w2, 0, 0, w2};
          if (oinfo) {
            for (int t = 0; t < 4; ++t) { L1[t] = w1 * oinfo[4 * k1 + t]; L2[t] = w2 * oinfo[4 * k2 + t]; }
          }
          double LJc[12], LJp2[6], LJp1[6];
          for (int c = 0; c < 6; ++c) {
            LJc[c] = L2[0] * Jc2[c] + L2[1] * Jc2[6 + c];
            LJc[6 + c] = L2[2] * Jc2[c] + L2[3] * Jc2[6 + c];
          }
          for (int c = 0; c < 3; ++c) {
            LJp2[c] = L2[0] * Jp2[c] + L2[1] * Jp2[3 + c];
            LJp2[3 + c] = L2[2] * Jp2[c] + L2[3] * Jp2[3 + c];
            LJp1[c] = L1[0] * Jp1[c] + L1[1] * Jp1[3 + c];
            LJp1[3 + c] = L1[2] * Jp1[c] + L1[3] * Jp1[3 + c];
          }
          double A[36], B[18], V[9];
          for (int a = 0; a < 6; ++a) {
            for (int b = 0; b < 6; ++b) A[6 * a + b] = Jc2[a] * LJc[b] + Jc2[6 + a] * LJc[6 + b];
            for (int b = 0; b < 3; ++b) B[3 * a + b] = Jc2[a] * LJp2[b] + Jc2[6 + a] * LJp2[3 + b];
          }
          for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
              V[3 * a + b] = (Jp2[a] * LJp2[b] + Jp2[3 + a] * LJp2[3 + b]) + (Jp1[a] * LJp1[b] + Jp1[3 + a] * LJp1[3 + b]);
          const double c00 = V[4] * V[8] - V[5] * V[7], c01 = V[5] * V[6] - V[3] * V[8], c02 = V[3] * V[7] - V[4] * V[6];
          const double det = V[0] * c00 + V[1] * c01 + V[2] * c02;
          if (pf && det > 0.0) {
            const double id = 1.0 / det;
            const double Vi[9] = {c00 * id, (V[2] * V[7] - V[1] * V[8]) * id, (V[1] * V[5] - V[2] * V[4]) * id,
                                  c01 * id, (V[0] * V[8] - V[2] * V[6]) * id, (V[2] * V[3] - V[0] * V[5]) * id,
                                  c02 * id, (V[1] * V[6] - V[0] * V[7]) * id, (V[0] * V[4] - V[1] * V[3]) * id};
            double BV[18];
            for (int a = 0; a < 6; ++a)
              for (int b = 0; b < 3; ++b) BV[3 * a + b] = B[3 * a] * Vi[b] + B[3 * a + 1] * Vi[3 + b] + B[3 * a + 2] * Vi[6 + b];
            for (int a = 0; a < 6; ++a)
              for (int b = 0; b < 6; ++b) A[6 * a + b] -= BV[3 * a] * B[3 * b] + BV[3 * a + 1] * B[3 * b + 1] + BV[3 * a + 2] * B[3 * b + 2];
          }
          for (int t = 0; t < 36; ++t) Lam[t] += A[t];
        }
        edge_first[n_edges] = ci;
        edge_second[n_edges] = cj;
        if (edge_shared) edge_shared[n_edges] = (int32_t)(b1 - b0);
        for (int t = 0; t < 36; ++t) edge_info[36 * (size_t)n_edges + t] = Lam[t];
      }
      ++n_edges;
    }
    b0 = b1;
  }
  free(ent);
  free(cur);
  free(plist);
  free(pstart);
  return n_edges;
}
