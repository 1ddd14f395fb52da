/*
 * pg_oracle.c — CPU restatement of the pose-graph / alignment side of GSLAM::Optimizer.  TEST INFRASTRUCTURE ONLY
 * (see oracle/README): loaded by tests/ and bench.py's CPU legs, never by the product.
 *
 * What the reference pins and what it does not:
 *   PINNED   the Lie algebra: SIM3::exp / log / operator* (GSLAM/core/SIM3.h:114-270), SE3::log / exp / inverse
 *            (GSLAM/core/SE3.h:100-103,205-287), SO3 products (SO3.h:489-509).  oracle/_ref compiles those headers and
 *            tests/golden/sim3_reference.npz holds their outputs (tools/gen_golden.py); this file restates them with
 *            series expansions where the reference divides 0 / 0 (theta -> 0, sigma -> 0).
 *   UNPINNED the solvers.  GSLAM/core/Optimizer.h:127-148,162-167 only DEFINES the pose-graph data (SE3Edge: measurement
 *            SE3_12 := SE3_1^-1 SE3_2 with a 6x6 information; SIM3Edge likewise with 7x7; GPSEdge: SE3_gps := SE3_frame)
 *            and :210-225 the signatures of optimizeICP / fitSim3; every implementation lives in un-vendored plugins
 *            (CMakeLists.txt:44, commented out).  Specified here, cross-checked in tests/test_pg_oracle.py against
 *            scipy.optimize.least_squares (same residuals, independent Jacobians) and numpy's SVD-based Umeyama.
 *
 * Specification (DESIGN.md section 4g repeats it):
 *   state      keyframe i = SIM3 T_wc as [qx qy qz qw tx ty tz s] (GSLAM field order), dof mask = KeyFrameEstimzationDOF
 *              (X Y Z RX RY RZ SCALE = bits 0..6, Optimizer.h:70-84)
 *   update     S <- S * SIM3::exp(delta), delta = [v(3) w(3) sigma]; masked components of delta are zero; the
 *              quaternion is renormalised (same right-multiplicative convention as the bundle adjustment, SE3.h:120-123)
 *   residuals  SE3 edge   r = SE3::log( M^-1 * (T_i^-1 * T_j) )   6-vector, T = (R, t) of the keyframe (scale ignored)
 *              SIM3 edge  r = SIM3::log( M^-1 * (S_i^-1 * S_j) )  7-vector
 *              GPS edge   r = SE3::log( M^-1 * T_i )              6-vector
 *   cost       1/2 sum r^T Lambda r (Lambda = the edge's information, identity when absent); no robust kernel (the
 *              Huber threshold of OptimzeConfig is a PROJECTION-error threshold, Optimizer.h:176)
 *   Jacobians  central differences of r along each component of delta, h = 1e-6 (what g2o does for its Sim3 edges);
 *              masked components get a zero column
 *   solver     Levenberg-Marquardt on the dense normal equations H = sum J^T Lambda J (7 n_frames square), the same
 *              trust-region strategy as the bundle adjustment (ba_oracle.c: damping clamp(H_ii, 1e-6, 1e32) / radius,
 *              rho test, radius / decrease schedule, the three termination rules)
 *   alignment  dst ~ s R src + t over n 3-D correspondences: Horn's closed form (quaternion of the largest eigenvalue of
 *              the 4x4 N matrix, scale sqrt(sum |b|^2 / sum |a|^2), or 1 when the scale is not a degree of freedom),
 *              then the information J^T J of the residual dst - S src at the solution for the right-multiplicative delta.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PG_MAX_TRACE 512

typedef struct {
  double huber_delta; /* unused here */
  int32_t max_iterations;
  double initial_radius, function_tolerance, gradient_tolerance, min_relative_decrease;
  int32_t verbose, deterministic;
} pg_options; /* layout of gh_ba_options / oracle_ba_options */

typedef struct {
  int32_t iterations, accepted, termination;
  double initial_cost, final_cost, solve_ms_total, total_ms;
  int32_t trace_len;
  double trace_cost[PG_MAX_TRACE], trace_radius[PG_MAX_TRACE];
  uint8_t trace_accepted[PG_MAX_TRACE];
} pg_summary;

int oracle_potrf(double* A, int n, int threads); /* ba_oracle.c */
void oracle_potrs(const double* L, int n, double* b);

/* ---------------------------------------------------------------- quaternion / SIM3 algebra, sim = qx qy qz qw tx ty tz s */
static void q_rot(const double* q, const double* p, double* o) { /* SO3.h:497-509 */
  double uvx = q[1] * p[2] - q[2] * p[1], uvy = q[2] * p[0] - q[0] * p[2], uvz = q[0] * p[1] - q[1] * p[0];
  uvx += uvx; uvy += uvy; uvz += uvz;
  o[0] = p[0] + q[3] * uvx + (q[1] * uvz - q[2] * uvy);
  o[1] = p[1] + q[3] * uvy + (q[2] * uvx - q[0] * uvz);
  o[2] = p[2] + q[3] * uvz + (q[0] * uvy - q[1] * uvx);
}
static void q_mul(const double* a, const double* b, double* o) { /* SO3.h:489-495 */
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
static void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

/* SIM3::operator* (SIM3.h:114-119): (R_a R_b, t_a + R_a (s_a t_b), s_a s_b) */
void oracle_sim3_mul(const double* a, const double* b, double* o) {
  double q[4], t[3], st[3] = {a[7] * b[4], a[7] * b[5], a[7] * b[6]};
  q_mul(a, b, q);
  q_rot(a, st, t);
  memcpy(o, q, 32);
  for (int e = 0; e < 3; ++e) o[4 + e] = a[4 + e] + t[e];
  o[7] = a[7] * b[7];
}

/* SIM3::inv (SIM3.h:126-131; the reference's body calls a non-existent SO3::inv and does not compile, the formula is
 * the one it states): (R^T, -(1/s) R^T t, 1/s) */
void oracle_sim3_inv(const double* a, double* o) {
  double qc[4] = {-a[0], -a[1], -a[2], a[3]}, t[3];
  q_rot(qc, a + 4, t);
  const double is = 1.0 / a[7];
  memcpy(o, qc, 32);
  for (int e = 0; e < 3; ++e) o[4 + e] = -is * t[e];
  o[7] = is;
}

/* rotation part shared by SE3::log and SIM3::log (SE3.h:212-245 / SIM3.h:196-224): r = theta * axis from the quaternion */
static double rot_log(const double* q, double* r) {
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  double A_inv;
  if (n < 1e-10) {
    const double w2 = q[3] * q[3];
    A_inv = 2.0 / q[3] - 2.0 * (1.0 - w2) / (q[3] * w2);
  } else if (fabs(q[3]) < 1e-10) {
    A_inv = (q[3] > 0 ? 3.14159265358979323846 : -3.14159265358979323846) / n;
  } else {
    A_inv = 2.0 * atan(n / q[3]) / n;
  }
  r[0] = q[0] * A_inv;
  r[1] = q[1] * A_inv;
  r[2] = q[2] * A_inv;
  return A_inv * n; /* theta (signed like the reference's) */
}

/* Coefficients of W = C I + A [r]x + B [r]x^2 in t = W p (SIM3.h:158-183; Sophus' Sim3 formulas):
 *   C = (s - 1) / sigma,  A = (a sigma + (1 - b) theta) / (theta (theta^2 + sigma^2)),
 *   B = (C - ((b - 1) sigma + a theta) / (theta^2 + sigma^2)) / theta^2,   a = s sin theta, b = s cos theta, s = e^sigma.
 * theta -> 0: A = int_0^1 tau e^(sigma tau) dtau = ((sigma - 1) s + 1) / sigma^2 and B = 1/2 int tau^2 e^(sigma tau) =
 * ((sigma^2 / 2 - sigma + 1) s - 1) / sigma^3 (the reference's B lacks the "- 1": harmless, B multiplies an O(theta^2)
 * term there); sigma -> 0 as well: the Taylor series of those integrals.  C through expm1 (no cancellation). */
static void sim3_abc(double theta, double sigma, double* A, double* B, double* C) {
  const double th = fabs(theta), th2 = th * th, scale = exp(sigma);
  *C = fabs(sigma) < 1e-12 ? 1.0 + 0.5 * sigma : expm1(sigma) / sigma;
  if (th < 1e-5) {
    if (fabs(sigma) < 1e-3) {
      *A = 0.5 + sigma * (1.0 / 3.0 + sigma * (1.0 / 8.0 + sigma / 30.0));
      *B = 1.0 / 6.0 + sigma * (1.0 / 8.0 + sigma * (1.0 / 20.0 + sigma / 72.0));
    } else {
      const double s2 = sigma * sigma;
      *A = ((sigma - 1.0) * scale + 1.0) / s2;
      *B = ((0.5 * s2 - sigma + 1.0) * scale - 1.0) / (s2 * sigma);
    }
    /* first correction in theta^2 (keeps exp / log inverse of each other to 1e-15 for theta < 1e-5) is below 1e-11 */
    return;
  }
  const double a = scale * sin(th), b = scale * cos(th), c = th2 + sigma * sigma;
  *A = (a * sigma + (1.0 - b) * th) / (th * c);
  *B = (*C - ((b - 1.0) * sigma + a * th) / c) / th2;
}

/* SIM3::exp (SIM3.h:133-188), mu = [p(3), r(3), sigma] */
void oracle_sim3_exp(const double* mu, double* S) {
  const double* p = mu;
  const double* r = mu + 3;
  const double th2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2], th = sqrt(th2);
  double imag, real;
  if (th < 1e-5) {
    const double th4 = th2 * th2;
    imag = 0.5 - th2 / 48.0 + th4 / 3840.0;
    real = 1.0 - th2 / 8.0 + th4 / 384.0;
  } else {
    imag = sin(0.5 * th) / th;
    real = cos(0.5 * th);
  }
  double A, B, C;
  sim3_abc(th, mu[6], &A, &B, &C);
  S[0] = imag * r[0];
  S[1] = imag * r[1];
  S[2] = imag * r[2];
  S[3] = real;
  double c1[3], c2[3];
  cross3(r, p, c1);
  cross3(r, c1, c2);
  for (int e = 0; e < 3; ++e) S[4 + e] = A * c1[e] + B * c2[e] + C * p[e];
  S[7] = exp(mu[6]);
}

/* SIM3::log (SIM3.h:190-270): p = W^-1 t.  On the plane orthogonal to r, W acts as the complex number x + i y with
 * x = C - B theta^2, y = A theta, and as C along r, so W^-1 = c I + a [r]x + b [r]x^2 with
 *   c = 1 / C,  a = -A / (x^2 + y^2),  b = (A^2 - B x) / (C (x^2 + y^2))
 * -- algebraically the reference's closed forms (:232-262), finite at theta = 0 and sigma = 0 without a special case. */
void oracle_sim3_log(const double* S, double* mu) {
  double r[3];
  const double theta = rot_log(S, r);
  const double sigma = log(S[7]);
  double A, B, C;
  sim3_abc(theta, sigma, &A, &B, &C);
  const double th2 = theta * theta, x = C - B * th2, d = x * x + A * A * th2;
  const double ci = 1.0 / C, ai = -A / d, bi = (A * A - B * x) / (C * d);
  const double* t = S + 4;
  double c1[3], c2[3];
  cross3(r, t, c1);
  cross3(r, c1, c2);
  for (int e = 0; e < 3; ++e) {
    mu[e] = ci * t[e] + ai * c1[e] + bi * c2[e];
    mu[3 + e] = r[e];
  }
  mu[6] = sigma;
}

/* SE3::log (SE3.h:205-255) = the scale-1 case: xi = [p(3), r(3)] */
void oracle_se3_log(const double* T /* qx qy qz qw tx ty tz */, double* xi) {
  const double S[8] = {T[0], T[1], T[2], T[3], T[4], T[5], T[6], 1.0};
  double mu[7];
  oracle_sim3_log(S, mu);
  memcpy(xi, mu, 48);
}

/* S * exp(delta), quaternion renormalised */
void oracle_sim3_retract(const double* S, const double* delta, double* out) {
  double E[8];
  oracle_sim3_exp(delta, E);
  oracle_sim3_mul(S, E, out);
  const double n = 1.0 / sqrt(out[0] * out[0] + out[1] * out[1] + out[2] * out[2] + out[3] * out[3]);
  for (int e = 0; e < 4; ++e) out[e] *= n;
}

/* ---------------------------------------------------------------- residuals */
/* type 0 SE3 edge (meas 7), 1 SIM3 edge (meas 8), 2 GPS edge (meas 7; Sj unused).  Returns the residual dimension. */
static int edge_residual(int type, const double* Si, const double* Sj, const double* meas, double* r) {
  if (type == 1) {
    double Mi[8], Sii[8], E1[8], E2[8];
    oracle_sim3_inv(meas, Mi);
    oracle_sim3_inv(Si, Sii);
    oracle_sim3_mul(Sii, Sj, E1);
    oracle_sim3_mul(Mi, E1, E2);
    oracle_sim3_log(E2, r);
    return 7;
  }
  double M[8] = {meas[0], meas[1], meas[2], meas[3], meas[4], meas[5], meas[6], 1.0}, Mi[8];
  double Ti[8] = {Si[0], Si[1], Si[2], Si[3], Si[4], Si[5], Si[6], 1.0}, E2[8];
  oracle_sim3_inv(M, Mi);
  if (type == 0) {
    double Tj[8] = {Sj[0], Sj[1], Sj[2], Sj[3], Sj[4], Sj[5], Sj[6], 1.0}, Tii[8], E1[8];
    oracle_sim3_inv(Ti, Tii);
    oracle_sim3_mul(Tii, Tj, E1);
    oracle_sim3_mul(Mi, E1, E2);
  } else {
    oracle_sim3_mul(Mi, Ti, E2);
  }
  oracle_se3_log(E2, r);
  return 6;
}

#define PG_FD_STEP 1e-6
/* J (dim x 7, row-major 7 x 7 storage) of the residual w.r.t. the right-multiplicative delta of endpoint `which` */
static void edge_jacobian(int type, const double* Si, const double* Sj, const double* meas, int which, int dof, double* J) {
  memset(J, 0, 49 * 8);
  for (int k = 0; k < 7; ++k) {
    if (!((dof >> k) & 1)) continue;
    double dp[7] = {0, 0, 0, 0, 0, 0, 0}, Sp[8], Sm[8], rp[7], rm[7];
    dp[k] = PG_FD_STEP;
    oracle_sim3_retract(which == 0 ? Si : Sj, dp, Sp);
    dp[k] = -PG_FD_STEP;
    oracle_sim3_retract(which == 0 ? Si : Sj, dp, Sm);
    const int dim = edge_residual(type, which == 0 ? Sp : Si, which == 0 ? Sj : Sp, meas, rp);
    edge_residual(type, which == 0 ? Sm : Si, which == 0 ? Sj : Sm, meas, rm);
    for (int a = 0; a < dim; ++a) J[7 * a + k] = (rp[a] - rm[a]) / (2.0 * PG_FD_STEP);
  }
}

typedef struct {
  int n_frames, n_edges;
  const int32_t* dof;
  const int32_t *etype, *ei, *ej;
  const double* meas; /* n_edges x 8 */
  const double* info; /* n_edges x 49 or NULL */
} pg_ctx;

static void edge_info(const pg_ctx* c, int e, int dim, double* L) {
  if (c->info) {
    memcpy(L, c->info + 49 * (size_t)e, 49 * 8);
    return;
  }
  memset(L, 0, 49 * 8);
  for (int a = 0; a < dim; ++a) L[7 * a + a] = 1.0;
}

static double pg_cost(const pg_ctx* c, const double* S) {
  double cost = 0;
  for (int e = 0; e < c->n_edges; ++e) {
    double r[7], L[49];
    const int i = c->ei[e], j = c->ej[e];
    const int dim = edge_residual(c->etype[e], S + 8 * i, j >= 0 ? S + 8 * j : S + 8 * i, c->meas + 8 * (size_t)e, r);
    edge_info(c, e, dim, L);
    double q = 0;
    for (int a = 0; a < dim; ++a) {
      double Lr = 0;
      for (int b = 0; b < dim; ++b) Lr += L[7 * a + b] * r[b];
      q += r[a] * Lr;
    }
    cost += 0.5 * q;
  }
  return cost;
}

static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* frames: n_frames x 8 in/out.  Edges as parallel arrays (type, i, j (-1 for GPS), meas 8, info 49 row-major with the
 * dim x dim block in the top-left corner, or NULL = identity). */
int oracle_pg_solve(int n_frames, double* frames, const int32_t* dof, int n_edges, const int32_t* etype, const int32_t* ei,
                    const int32_t* ej, const double* meas, const double* info, const pg_options* opt, pg_summary* sum,
                    int threads) {
  pg_ctx c = {n_frames, n_edges, dof, etype, ei, ej, meas, info};
  const int n = 7 * n_frames;
  double* H = (double*)malloc((size_t)n * n * 8);
  double* Hd = (double*)malloc((size_t)n * n * 8);
  double* g = (double*)malloc((size_t)n * 8);
  double* d = (double*)malloc((size_t)n * 8);
  double* Snew = (double*)malloc((size_t)n_frames * 8 * 8);
  memset(sum, 0, sizeof(*sum));
  double radius = opt->initial_radius, decrease = 2.0;
  double cost = pg_cost(&c, frames);
  sum->initial_cost = cost;
  int need_lin = 1, term = 0, it = 0;
  for (it = 0; it < opt->max_iterations; ++it) {
    if (need_lin) {
      memset(H, 0, (size_t)n * n * 8);
      memset(g, 0, (size_t)n * 8);
      for (int e = 0; e < n_edges; ++e) {
        const int i = ei[e], j = ej[e];
        const double* Si = frames + 8 * i;
        const double* Sj = j >= 0 ? frames + 8 * j : Si;
        double r[7], L[49], Ji[49], Jj[49], LJi[49], LJj[49], Lr[7];
        const int dim = edge_residual(etype[e], Si, Sj, meas + 8 * (size_t)e, r);
        edge_info(&c, e, dim, L);
        edge_jacobian(etype[e], Si, Sj, meas + 8 * (size_t)e, 0, dof[i], Ji);
        if (j >= 0) edge_jacobian(etype[e], Si, Sj, meas + 8 * (size_t)e, 1, dof[j], Jj);
        else memset(Jj, 0, sizeof(Jj));
        for (int a = 0; a < dim; ++a) {
          double s = 0;
          for (int b = 0; b < dim; ++b) s += L[7 * a + b] * r[b];
          Lr[a] = s;
          for (int k = 0; k < 7; ++k) {
            double si = 0, sj = 0;
            for (int b = 0; b < dim; ++b) {
              si += L[7 * a + b] * Ji[7 * b + k];
              sj += L[7 * a + b] * Jj[7 * b + k];
            }
            LJi[7 * a + k] = si;
            LJj[7 * a + k] = sj;
          }
        }
        for (int p = 0; p < 7; ++p) {
          double gi = 0, gj = 0;
          for (int a = 0; a < dim; ++a) {
            gi += Ji[7 * a + p] * Lr[a];
            gj += Jj[7 * a + p] * Lr[a];
          }
          g[7 * i + p] += gi;
          if (j >= 0) g[7 * j + p] += gj;
          for (int q = 0; q < 7; ++q) {
            double hii = 0, hjj = 0, hji = 0;
            for (int a = 0; a < dim; ++a) {
              hii += Ji[7 * a + p] * LJi[7 * a + q];
              hjj += Jj[7 * a + p] * LJj[7 * a + q];
              hji += Jj[7 * a + p] * LJi[7 * a + q]; /* block (j, i) */
            }
            H[(size_t)(7 * i + q) * n + 7 * i + p] += hii; /* column-major */
            if (j >= 0) {
              H[(size_t)(7 * j + q) * n + 7 * j + p] += hjj;
              H[(size_t)(7 * i + q) * n + 7 * j + p] += hji;
              H[(size_t)(7 * j + p) * n + 7 * i + q] += hji; /* the symmetric partner */
            }
          }
        }
      }
      double gmax = 0;
      for (int k = 0; k < n; ++k) gmax = fmax(gmax, fabs(g[k]));
      if (gmax <= opt->gradient_tolerance) { term = 2; break; }
      need_lin = 0;
    }
    memcpy(Hd, H, (size_t)n * n * 8);
    for (int k = 0; k < n; ++k) {
      Hd[(size_t)k * n + k] += clampd(H[(size_t)k * n + k], 1e-6, 1e32) / radius;
      d[k] = -g[k];
    }
    int ok = oracle_potrf(Hd, n, threads) == 0;
    double new_cost = cost, model = 0, rho = -1;
    if (ok) {
      oracle_potrs(Hd, n, d);
      /* model decrease -(g^T d + 1/2 d^T H d) with the UNDAMPED H */
      for (int a = 0; a < n; ++a) {
        double hd = 0;
        for (int b = 0; b < n; ++b) hd += H[(size_t)b * n + a] * d[b];
        model -= d[a] * (g[a] + 0.5 * hd);
      }
      for (int f = 0; f < n_frames; ++f) {
        if ((dof[f] & 127) == 0) memcpy(Snew + 8 * f, frames + 8 * f, 64);
        else oracle_sim3_retract(frames + 8 * f, d + 7 * f, Snew + 8 * f);
      }
      new_cost = pg_cost(&c, Snew);
      rho = model > 0 ? (cost - new_cost) / model : -1;
      if (!(new_cost == new_cost)) rho = -1;
    }
    const int acc = ok && rho > opt->min_relative_decrease;
    if (sum->trace_len < PG_MAX_TRACE) {
      sum->trace_cost[sum->trace_len] = new_cost;
      sum->trace_radius[sum->trace_len] = radius;
      sum->trace_accepted[sum->trace_len] = (uint8_t)acc;
      sum->trace_len++;
    }
    if (acc) {
      const double dcost = cost - new_cost;
      memcpy(frames, Snew, (size_t)n_frames * 64);
      const double t = 2.0 * rho - 1.0;
      radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
      if (radius > 1e16) radius = 1e16;
      decrease = 2.0;
      sum->accepted++;
      need_lin = 1;
      const double prev = cost;
      cost = new_cost;
      if (fabs(dcost) <= opt->function_tolerance * prev) { term = 1; ++it; break; }
    } else {
      radius = radius / decrease;
      decrease *= 2.0;
      if (radius < 1e-32) { term = 3; ++it; break; }
    }
  }
  sum->iterations = it;
  sum->termination = term;
  sum->final_cost = cost;
  free(H); free(Hd); free(g); free(d); free(Snew);
  return term == 3 ? 1 : 0;
}

double oracle_pg_cost(int n_frames, const double* frames, int n_edges, const int32_t* etype, const int32_t* ei,
                      const int32_t* ej, const double* meas, const double* info) {
  pg_ctx c = {n_frames, n_edges, NULL, etype, ei, ej, meas, info};
  return pg_cost(&c, frames);
}

int oracle_pg_edge_residual(int type, const double* Si, const double* Sj, const double* meas, double* r) {
  return edge_residual(type, Si, Sj, meas, r);
}

/* ---------------------------------------------------------------- 3-D alignment (optimizeICP / fitSim3) */
static void jacobi4(double a[4][4], double v[4][4]) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) v[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 16; ++sweep)
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        const double apq = a[p][q];
        if (!(fabs(apq) > 1e-300)) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < 4; ++k) {
          const double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - sn * akq;
          a[k][q] = sn * akp + c * akq;
        }
        for (int k = 0; k < 4; ++k) {
          const double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - sn * aqk;
          a[q][k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < 4; ++k) {
          const double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = c * vkp - sn * vkq;
          v[k][q] = sn * vkp + c * vkq;
        }
      }
}

/* From the 13 sums {sum a (3), sum b (3), sum a b^T (9 row-major), sum |a|^2, sum |b|^2} of n correspondences:
 * dst ~ s R src + t.  with_scale = 0 fixes s = 1.  Returns 0 for a degenerate set. */
int oracle_align_from_sums(const double* sums, int n, int with_scale, double* out8) {
  if (n < 3) return 0;
  const double inv = 1.0 / n;
  double ca[3], cb[3], M[3][3];
  for (int e = 0; e < 3; ++e) {
    ca[e] = sums[e] * inv;
    cb[e] = sums[3 + e] * inv;
  }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) M[r][c] = sums[6 + 3 * r + c] - n * ca[r] * cb[c];
  const double na = sums[15] - n * (ca[0] * ca[0] + ca[1] * ca[1] + ca[2] * ca[2]);
  const double nb = sums[16] - n * (cb[0] * cb[0] + cb[1] * cb[1] + cb[2] * cb[2]);
  if (!(na > 1e-300) || !(nb > 1e-300)) return 0;
  double N[4][4] = {{M[0][0] + M[1][1] + M[2][2], M[1][2] - M[2][1], M[2][0] - M[0][2], M[0][1] - M[1][0]},
                    {0, M[0][0] - M[1][1] - M[2][2], M[0][1] + M[1][0], M[2][0] + M[0][2]},
                    {0, 0, -M[0][0] + M[1][1] - M[2][2], M[1][2] + M[2][1]},
                    {0, 0, 0, -M[0][0] - M[1][1] + M[2][2]}};
  for (int r = 1; r < 4; ++r)
    for (int c = 0; c < r; ++c) N[r][c] = N[c][r];
  double V[4][4];
  jacobi4(N, V);
  int best = 0;
  for (int k = 1; k < 4; ++k)
    if (N[k][k] > N[best][best]) best = k;
  double qw = V[0][best], qx = V[1][best], qy = V[2][best], qz = V[3][best];
  const double qn = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
  if (!(qn > 1e-300)) return 0;
  if (qw < 0) { qw = -qw; qx = -qx; qy = -qy; qz = -qz; }
  qw /= qn; qx /= qn; qy /= qn; qz /= qn;
  const double sc = with_scale ? sqrt(nb / na) : 1.0;
  const double q[4] = {qx, qy, qz, qw};
  double Rca[3];
  q_rot(q, ca, Rca);
  out8[0] = qx; out8[1] = qy; out8[2] = qz; out8[3] = qw;
  for (int e = 0; e < 3; ++e) out8[4 + e] = cb[e] - sc * Rca[e];
  out8[7] = sc;
  return 1;
}

/* dst ~ S src: closed form + the 7x7 information J^T J of r_k = dst_k - S src_k w.r.t. the right-multiplicative delta
 * (S exp(delta) src = S (src + v + w x src + sigma src) to first order: J_k = -s R [I | -[src]x | src], masked by dof).
 * Returns the sum of squared residuals through *ssq. */
int oracle_align_sim3(const double* src, const double* dst, int n, int dof, double* out8, double* info49, double* ssq) {
  double sums[17];
  memset(sums, 0, sizeof(sums));
  for (int k = 0; k < n; ++k) {
    const double* a = src + 3 * k;
    const double* b = dst + 3 * k;
    for (int e = 0; e < 3; ++e) {
      sums[e] += a[e];
      sums[3 + e] += b[e];
      for (int f = 0; f < 3; ++f) sums[6 + 3 * e + f] += a[e] * b[f];
    }
    sums[15] += a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    sums[16] += b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
  }
  if (!oracle_align_from_sums(sums, n, (dof >> 6) & 1, out8)) return 0;
  if (info49) memset(info49, 0, 49 * 8);
  double acc = 0;
  for (int k = 0; k < n; ++k) {
    const double* a = src + 3 * k;
    double Ra[3];
    q_rot(out8, a, Ra);
    double r[3];
    for (int e = 0; e < 3; ++e) {
      r[e] = dst[3 * k + e] - (out8[7] * Ra[e] + out8[4 + e]);
      acc += r[e] * r[e];
    }
    if (info49) {
      /* columns of D = [I | -[a]x | a] (3 x 7), J = -s R D: J^T J = s^2 D^T D (R orthonormal) */
      const double D[3][7] = {{1, 0, 0, 0, a[2], -a[1], a[0]}, {0, 1, 0, -a[2], 0, a[0], a[1]}, {0, 0, 1, a[1], -a[0], 0, a[2]}};
      for (int p = 0; p < 7; ++p)
        for (int q2 = 0; q2 < 7; ++q2) {
          if (!((dof >> p) & 1) || !((dof >> q2) & 1)) continue;
          double s = 0;
          for (int e = 0; e < 3; ++e) s += D[e][p] * D[e][q2];
          info49[7 * p + q2] += out8[7] * out8[7] * s;
        }
    }
  }
  if (ssq) *ssq = acc;
  return 1;
}
