/*
 * graph_oracle.c — CPU restatement of the GENERAL BundleGraph solve of GSLAM::Optimizer::optimize: SIM3 keyframes,
 * pose-graph edges (se3Graph / sim3Graph / gpsGraph), XYZ map points AND inverse-depth points with their observations,
 * all in one graph (GSLAM/core/Optimizer.h:102-172).  TEST INFRASTRUCTURE ONLY (see oracle/README): loaded by tests/,
 * never by the product.
 *
 * PARITY UNPINNED.  The reference defines the containers only: InvDepthEstimation {frameId, anchor, estimation =
 * [idepth, sigma], dof} (:106-111), BundleEdge {pointId, frameId, measurement, information 2x2} (:121-125),
 * invDepthObserves / mappointObserves (:160-161); no implementation of optimize() exists in the tree (the optimizer
 * plugins are un-vendored, CMakeLists.txt:44).  Specified here, cross-checked in tests/test_graph_oracle.py against
 * scipy.optimize.least_squares (same residuals, independent finite-difference Jacobians) and against ba_oracle.c on
 * graphs both can express.
 *
 * Specification (DESIGN.md section 4g repeats it):
 *   keyframe j     SIM3 S_j = (R_j, t_j, s_j): camera -> world, X_w = s_j R_j X_c + t_j; update S <- S * SIM3::exp(delta),
 *                  delta = [v w sigma] masked by KeyFrameEstimzationDOF (exactly pg_oracle.c)
 *   XYZ point      world X, additive update, MapPointEstimation.second = false keeps it fixed
 *   inverse depth  host keyframe h, anchor a in the host camera (pinhole: (x, y, 1)), idepth rho > 0: X_c(h) = a / rho;
 *                  additive update of rho (floored at 1e-9) when dof has UPDATE_ID_IDEPTH; sigma is carried, never touched
 *   observation    of a landmark in keyframe j, measurement m = (m_x, m_y, 1), PINHOLE projection:
 *                    XYZ:        Y = R_j^T (X - t_j)
 *                    inv. depth: Y = R_j^T (s_h R_h a + rho (t_h - t_j))        (= rho s_j X_c(j): the ratio below is X_c's)
 *                    r = (Y_x / Y_z - m_x, Y_y / Y_z - m_y);  dropped while Y_z <= 1e-9 (not in front of the camera)
 *                  s = r^T Lambda r (Lambda = the edge's 2x2 information, identity when absent), Huber on sqrt(s) with
 *                  OptimzeConfig::projectErrorHuberThreshold: cost 1/2 rho(s), rho(s) = s or 2 h sqrt(s) - h^2; IRLS
 *                  weight w = 1 or h / sqrt(s) fixed at the linearisation point (ba_oracle.c's convention)
 *                  An observation of an inverse-depth point in its own host frame is constant (a_xy / a_z - m): it adds
 *                  to the cost and has no Jacobian.
 *   sphere         CameraProjectionType PROJECTION_SPHERE ("||x,y,z|| = 1", Optimizer.h:58-61): anchors and measurements
 *                  are unit bearings b; the inverse depth is an inverse RANGE (X_c(h) = a / rho, |a| = 1).  Residual = the
 *                  predicted bearing y = Y / |Y| in the tangent plane of the measured one: r = (e1 . y, e2 . y) with
 *                  e1 = normalise(b x k), e2 = b x e1, k = the coordinate axis b is least aligned with (ties: x, then y);
 *                  dropped while y . b <= 0 (opposite hemisphere).  Same information / Huber; the projection Jacobian
 *                  becomes P = E (I - y y^T) / |Y|, everything else is unchanged.
 *   Jacobians      analytic, first order in the right-multiplicative delta (R' = R (I + [w]x), t' = t + s R v,
 *                  s' = s (1 + sigma)), P = (1 / Y_z) [1 0 -u; 0 1 -v]:
 *                    dY/dv_j = -c s_j I   (c = 1 for XYZ, rho for inverse depth),  dY/dw_j = [Y]x,  dY/dsigma_j = 0
 *                    dY/dX = R_j^T;   dY/drho = R_j^T (t_h - t_j)
 *                    dY/dv_h = rho s_h R_j^T R_h,  dY/dw_h = -s_h R_j^T R_h [a]x,  dY/dsigma_h = s_h R_j^T R_h a
 *                  pose-graph edges: central differences as in pg_oracle.c
 *   intrinsics     BundleGraph::camera + cameraDOF (Optimizer.h:86-100,169-171: "Invalid camera indicates idea camera"):
 *                  with a camera c = (fx, fy, cx, cy, k1, k2, p1, p2, k3) the measurements are PIXELS and the residual is
 *                  r = Project_c(Y) - m with GSLAM's OpenCV model (GSLAM/core/Camera.h:386-407; the pinhole model :213-227
 *                  is k = p = 0):  x = Y_x / Y_z, y = Y_y / Y_z, r2 = x^2 + y^2, rad = 1 + k1 r2 + k2 r2^2 + k3 r2^3,
 *                    X1 = x rad + 2 p1 x y + p2 (r2 + 2 x^2),  Y1 = y rad + 2 p2 x y + p1 (r2 + 2 y^2),
 *                    U = cx + fx X1,  V = cy + fy Y1.
 *                  The parameters whose bit is set in intrinsics_free (bit i = parameter i of c; CameraEstimationDOF maps
 *                  FOCAL -> fx fy, CENTER -> cx cy, K1 K2 P1 P2 K3) are unknowns of the SAME Levenberg-Marquardt problem:
 *                  additive update, analytic Jacobian d(U, V)/dc, same damping rule, a 9-row block behind the keyframe
 *                  unknowns of the reduced system (the landmarks' Schur complement couples it to every keyframe).  The
 *                  projection Jacobian becomes P <- [d(U, V)/d(x, y)] P; Huber threshold and information are in pixels.
 *                  Pinhole projection only.
 *   solver         Levenberg-Marquardt, trust-region policy of ba_oracle.c / pg_oracle.c (damping clamp(H_kk, 1e-6, 1e32)
 *                  / radius on EVERY diagonal entry, keyframes and landmarks), landmarks eliminated by a Schur
 *                  complement (3x3 / 1x1 blocks), dense reduced system over the 7 n_frames keyframe unknowns, model
 *                  decrease -(g^T d + 1/2 d^T H d) with the undamped H summed edge by edge; a candidate that pushes a
 *                  previously valid observation behind its camera is rejected.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PG_MAX_TRACE 512

typedef struct {
  double huber_delta;
  int32_t max_iterations;
  double initial_radius, function_tolerance, gradient_tolerance, min_relative_decrease;
  int32_t verbose, deterministic;
} pg_options;

typedef struct {
  int32_t iterations, accepted, termination;
  double initial_cost, final_cost, solve_ms_total, total_ms;
  int32_t trace_len;
  double trace_cost[PG_MAX_TRACE], trace_radius[PG_MAX_TRACE];
  uint8_t trace_accepted[PG_MAX_TRACE];
} pg_summary;

typedef struct {
  int32_t n_frames;
  double* frames;          /* n_frames x 8, in / out */
  const int32_t* dof;      /* n_frames */
  int32_t n_edges;         /* pose-graph edges, flattened as oracle_pg_solve takes them */
  const int32_t *etype, *ei, *ej;
  const double* meas;      /* n_edges x 8 */
  const double* info;      /* n_edges x 49 or NULL */
  int32_t n_xyz;
  double* xyz;             /* n_xyz x 3, in / out */
  const uint8_t* xyz_free; /* NULL = all free */
  int32_t n_idp;
  const int32_t* idp_host;
  const double* idp_anchor; /* n_idp x 3 */
  double* idp_rho;          /* in / out */
  const uint8_t* idp_free;  /* NULL = all free */
  int32_t n_obs;
  const int32_t* obs_kind;  /* 0 XYZ point, 1 inverse-depth point */
  const int32_t* obs_point;
  const int32_t* obs_frame;
  const double* obs_xy;     /* n_obs x 2 */
  const double* obs_info;   /* n_obs x 4 or NULL */
  double huber;
  int32_t projection;         /* 0 pinhole (obs_xy), 1 sphere (obs_bearing) */
  const double* obs_bearing;  /* n_obs x 3 unit vectors, sphere only */
  double* intrinsics;         /* 9: fx fy cx cy k1 k2 p1 p2 k3, in / out; NULL = ideal camera (obs_xy on the z = 1 plane) */
  int32_t intrinsics_free;    /* bit i: parameter i is estimated */
} graph_problem;

/* pg_oracle.c / ba_oracle.c */
void oracle_sim3_retract(const double* S, const double* delta, double* out);
int oracle_pg_edge_residual(int type, const double* Si, const double* Sj, const double* meas, double* r);
int oracle_potrf(double* A, int n, int threads);
void oracle_potrs(const double* L, int n, double* b);

#define G_MIN_DEPTH 1e-9
#define G_FD_STEP 1e-6

static void q_rot(const double* q, const double* p, double* o) {
  double uvx = q[1] * p[2] - q[2] * p[1], uvy = q[2] * p[0] - q[0] * p[2], uvz = q[0] * p[1] - q[1] * p[0];
  uvx += uvx; uvy += uvy; uvz += uvz;
  o[0] = p[0] + q[3] * uvx + (q[1] * uvz - q[2] * uvy);
  o[1] = p[1] + q[3] * uvy + (q[2] * uvx - q[0] * uvz);
  o[2] = p[2] + q[3] * uvz + (q[0] * uvy - q[1] * uvx);
}
static void q_rot_inv(const double* q, const double* p, double* o) {
  const double qc[4] = {-q[0], -q[1], -q[2], q[3]};
  q_rot(qc, p, o);
}
static void q_matrix(const double* q, double* R) { /* row-major */
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

/* One observation.  kind 0: lm = X (3); kind 1: lm[0] = rho, host frame Sh, anchor a.  Returns 0 when the landmark is
 * not in front of the camera.  Jj / Jh: 2 x 7 (row-major), Jp: 2 x 3 (inverse depth: column 0).  same_host: the
 * observing frame IS the host (constant residual). */
static void tangent_basis(const double* b, double* e1, double* e2) {
  const double ax = fabs(b[0]), ay = fabs(b[1]), az = fabs(b[2]);
  double k[3] = {0, 0, 0};
  if (ax <= ay && ax <= az) k[0] = 1;
  else if (ay <= az) k[1] = 1;
  else k[2] = 1;
  e1[0] = b[1] * k[2] - b[2] * k[1];
  e1[1] = b[2] * k[0] - b[0] * k[2];
  e1[2] = b[0] * k[1] - b[1] * k[0];
  const double n = 1.0 / sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
  for (int e = 0; e < 3; ++e) e1[e] *= n;
  e2[0] = b[1] * e1[2] - b[2] * e1[1];
  e2[1] = b[2] * e1[0] - b[0] * e1[2];
  e2[2] = b[0] * e1[1] - b[1] * e1[0];
}

/* pixel coordinates of the normalised point (x, y) under c = fx fy cx cy k1 k2 p1 p2 k3 (GSLAM/core/Camera.h:396-406),
 * A = d(U, V)/d(x, y) row-major 2 x 2, Jc = d(U, V)/dc row-major 2 x 9 */
void oracle_cam_project(const double* c, double x, double y, double* UV, double* A, double* Jc) {
  const double fx = c[0], fy = c[1], k1 = c[4], k2 = c[5], p1 = c[6], p2 = c[7], k3 = c[8];
  const double x2 = x * x, y2 = y * y, r2 = x2 + y2, r4 = r2 * r2, r6 = r2 * r4, xy2 = x * y * 2.0;
  const double rad = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
  const double X1 = x * rad + xy2 * p1 + p2 * (r2 + 2.0 * x2);
  const double Y1 = y * rad + xy2 * p2 + p1 * (r2 + 2.0 * y2);
  UV[0] = c[2] + fx * X1;
  UV[1] = c[3] + fy * Y1;
  if (A) {
    const double radp = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4; /* d rad / d r2 */
    A[0] = fx * (rad + 2.0 * x2 * radp + 2.0 * p1 * y + 6.0 * p2 * x);
    A[1] = fx * (2.0 * x * y * radp + 2.0 * p1 * x + 2.0 * p2 * y);
    A[2] = fy * (2.0 * x * y * radp + 2.0 * p2 * y + 2.0 * p1 * x);
    A[3] = fy * (rad + 2.0 * y2 * radp + 2.0 * p2 * x + 6.0 * p1 * y);
  }
  if (Jc) {
    memset(Jc, 0, 18 * 8);
    Jc[0] = X1;            Jc[9 + 1] = Y1;
    Jc[2] = 1.0;           Jc[9 + 3] = 1.0;
    Jc[4] = fx * x * r2;   Jc[9 + 4] = fy * y * r2;
    Jc[5] = fx * x * r4;   Jc[9 + 5] = fy * y * r4;
    Jc[6] = fx * xy2;      Jc[9 + 6] = fy * (r2 + 2.0 * y2);
    Jc[7] = fx * (r2 + 2.0 * x2); Jc[9 + 7] = fy * xy2;
    Jc[8] = fx * x * r6;   Jc[9 + 8] = fy * y * r6;
  }
}

int oracle_graph_obs_cam(int kind, const double* Sj, int dof_j, const double* Sh, int dof_h, int same_host, const double* lm,
                         int lm_free, const double* anchor, const double* m, const double* info, double huber, double* r,
                         double* wgt, double* s_out, double* Jj, double* Jh, double* Jp, int projection, const double* cam,
                         int cam_free, double* Jc);

/* projection 0: m = (m_x, m_y) on the z = 1 plane; 1: m = unit bearing (3) */
int oracle_graph_obs(int kind, const double* Sj, int dof_j, const double* Sh, int dof_h, int same_host, const double* lm,
                     int lm_free, const double* anchor, const double* m, const double* info, double huber, double* r,
                     double* wgt, double* s_out, double* Jj, double* Jh, double* Jp, int projection) {
  return oracle_graph_obs_cam(kind, Sj, dof_j, Sh, dof_h, same_host, lm, lm_free, anchor, m, info, huber, r, wgt, s_out, Jj, Jh, Jp,
                              projection, NULL, 0, NULL);
}

/* cam != NULL (pinhole projection only): m in pixels, Jc = dr/dc (2 x 9, columns of fixed parameters zero) */
int oracle_graph_obs_cam(int kind, const double* Sj, int dof_j, const double* Sh, int dof_h, int same_host, const double* lm,
                         int lm_free, const double* anchor, const double* m, const double* info, double huber, double* r,
                         double* wgt, double* s_out, double* Jj, double* Jh, double* Jp, int projection, const double* cam,
                         int cam_free, double* Jc) {
  double Z[3], Y[3], Ra[3] = {0, 0, 0}, dth[3] = {0, 0, 0};
  if (kind == 0) {
    for (int e = 0; e < 3; ++e) Z[e] = lm[e] - Sj[4 + e];
  } else {
    q_rot(Sh, anchor, Ra);
    for (int e = 0; e < 3; ++e) {
      dth[e] = Sh[4 + e] - Sj[4 + e];
      Z[e] = Sh[7] * Ra[e] + lm[0] * dth[e];
    }
  }
  q_rot_inv(Sj, Z, Y);
  double P[6];
  if (projection == 0) {
    if (!(Y[2] > G_MIN_DEPTH)) return 0;
    const double iz = 1.0 / Y[2], u = Y[0] * iz, v = Y[1] * iz;
    P[0] = iz; P[1] = 0; P[2] = -u * iz; P[3] = 0; P[4] = iz; P[5] = -v * iz;
    if (cam) {
      double UV[2], A[4], Jcf[18];
      oracle_cam_project(cam, u, v, UV, A, Jcf);
      r[0] = UV[0] - m[0];
      r[1] = UV[1] - m[1];
      const double P0[6] = {P[0], P[1], P[2], P[3], P[4], P[5]};
      for (int e = 0; e < 3; ++e) {
        P[e] = A[0] * P0[e] + A[1] * P0[3 + e];
        P[3 + e] = A[2] * P0[e] + A[3] * P0[3 + e];
      }
      if (Jc)
        for (int a = 0; a < 2; ++a)
          for (int k = 0; k < 9; ++k) Jc[9 * a + k] = ((cam_free >> k) & 1) ? Jcf[9 * a + k] : 0.0;
    } else {
      r[0] = u - m[0];
      r[1] = v - m[1];
    }
  } else {
    const double nY = sqrt(Y[0] * Y[0] + Y[1] * Y[1] + Y[2] * Y[2]);
    if (!(nY > G_MIN_DEPTH)) return 0;
    const double in = 1.0 / nY, y[3] = {Y[0] * in, Y[1] * in, Y[2] * in};
    if (!(y[0] * m[0] + y[1] * m[1] + y[2] * m[2] > 0)) return 0;
    double e1[3], e2[3];
    tangent_basis(m, e1, e2);
    r[0] = e1[0] * y[0] + e1[1] * y[1] + e1[2] * y[2];
    r[1] = e2[0] * y[0] + e2[1] * y[1] + e2[2] * y[2];
    for (int e = 0; e < 3; ++e) {  /* E (I - y y^T) / |Y| */
      P[e] = (e1[e] - r[0] * y[e]) * in;
      P[3 + e] = (e2[e] - r[1] * y[e]) * in;
    }
  }
  double L00 = 1, L01 = 0, L10 = 0, L11 = 1;
  if (info) { L00 = info[0]; L01 = info[1]; L10 = info[2]; L11 = info[3]; }
  const double s = r[0] * (L00 * r[0] + L01 * r[1]) + r[1] * (L10 * r[0] + L11 * r[1]);
  double w = 1.0;
  if (huber > 0 && s > huber * huber) w = huber / sqrt(s);
  if (wgt) *wgt = w;
  if (s_out) *s_out = s;
  if (!Jj) return 1;
  memset(Jj, 0, 14 * 8);
  memset(Jh, 0, 14 * 8);
  memset(Jp, 0, 6 * 8);
  if (kind == 1 && same_host) return 1;
  const double c = kind == 0 ? 1.0 : lm[0];
  /* dY / d delta_j = [ -c s_j I | [Y]x | 0 ] */
  const double Dj[21] = {-c * Sj[7], 0, 0, 0, -Y[2], Y[1], 0,
                         0, -c * Sj[7], 0, Y[2], 0, -Y[0], 0,
                         0, 0, -c * Sj[7], -Y[1], Y[0], 0, 0};
  for (int a = 0; a < 2; ++a)
    for (int k = 0; k < 7; ++k) {
      double acc = 0;
      for (int e = 0; e < 3; ++e) acc += P[3 * a + e] * Dj[7 * e + k];
      Jj[7 * a + k] = ((dof_j >> k) & 1) ? acc : 0.0;
    }
  double Rj[9];
  q_matrix(Sj, Rj);
  if (kind == 0) {
    for (int a = 0; a < 2; ++a)
      for (int k = 0; k < 3; ++k) {
        double acc = 0;
        for (int e = 0; e < 3; ++e) acc += P[3 * a + e] * Rj[3 * k + e]; /* R_j^T[e][k] = R_j[k][e] */
        Jp[3 * a + k] = lm_free ? acc : 0.0;
      }
    return 1;
  }
  /* M = R_j^T R_h (3 x 3) */
  double Rh[9], M[9];
  q_matrix(Sh, Rh);
  for (int e = 0; e < 3; ++e)
    for (int f = 0; f < 3; ++f) {
      double acc = 0;
      for (int k = 0; k < 3; ++k) acc += Rj[3 * k + e] * Rh[3 * k + f];
      M[3 * e + f] = acc;
    }
  const double* a3 = anchor;
  /* -[a]x columns: (w x a) = -[a]x w  ->  d(R_h a) = R_h (w x a) */
  const double nax[9] = {0, a3[2], -a3[1], -a3[2], 0, a3[0], a3[1], -a3[0], 0};
  double Dh[21];
  for (int e = 0; e < 3; ++e) {
    for (int k = 0; k < 3; ++k) {
      Dh[7 * e + k] = lm[0] * Sh[7] * M[3 * e + k];
      double acc = 0;
      for (int f = 0; f < 3; ++f) acc += M[3 * e + f] * nax[3 * f + k];
      Dh[7 * e + 3 + k] = Sh[7] * acc;
    }
    Dh[7 * e + 6] = Sh[7] * (M[3 * e] * a3[0] + M[3 * e + 1] * a3[1] + M[3 * e + 2] * a3[2]);
  }
  for (int a = 0; a < 2; ++a)
    for (int k = 0; k < 7; ++k) {
      double acc = 0;
      for (int e = 0; e < 3; ++e) acc += P[3 * a + e] * Dh[7 * e + k];
      Jh[7 * a + k] = ((dof_h >> k) & 1) ? acc : 0.0;
    }
  double dr[3];
  q_rot_inv(Sj, dth, dr);
  for (int a = 0; a < 2; ++a) Jp[3 * a] = lm_free ? P[3 * a] * dr[0] + P[3 * a + 1] * dr[1] + P[3 * a + 2] * dr[2] : 0.0;
  return 1;
}

static double rho_huber(double s, double huber) {
  if (huber > 0 && s > huber * huber) return 2.0 * huber * sqrt(s) - huber * huber;
  return s;
}
static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

typedef struct {
  double r[2], w, L[4];
  int valid, fj, fh, lm, dp; /* dp = landmark dimension: 3 / 1 / 0 (fixed) */
  double Jj[14], Jh[14], Jp[6], Jc[18];
} obs_rec;

/* the (at most three) blocks of the reduced system an observation touches: observing keyframe, host keyframe (inverse-depth
 * point seen from another keyframe), intrinsics; J row-major 2 x width */
typedef struct {
  int ns, base[3], width[3];
  const double* J[3];
} obs_slots;
static obs_slots slots_of(const obs_rec* o, int nf, int with_cam) {
  obs_slots t;
  t.ns = 0;
  t.base[t.ns] = 7 * o->fj; t.width[t.ns] = 7; t.J[t.ns] = o->Jj; t.ns++;
  if (o->fh >= 0) { t.base[t.ns] = 7 * o->fh; t.width[t.ns] = 7; t.J[t.ns] = o->Jh; t.ns++; }
  if (with_cam) { t.base[t.ns] = 7 * nf; t.width[t.ns] = 9; t.J[t.ns] = o->Jc; t.ns++; }
  return t;
}

static int obs_eval(const graph_problem* g, int k, const double* frames, const double* xyz, const double* rho, const double* cam,
                    obs_rec* o, int with_j, double* s_out) {
  const int kind = g->obs_kind[k], p = g->obs_point[k], j = g->obs_frame[k];
  const int h = kind == 1 ? g->idp_host[p] : j;
  const double* info = g->obs_info ? g->obs_info + 4 * (size_t)k : NULL;
  const int lm_free = kind == 0 ? (g->xyz_free ? g->xyz_free[p] : 1) : (g->idp_free ? g->idp_free[p] : 1);
  const double* lm = kind == 0 ? xyz + 3 * (size_t)p : rho + p;
  double w = 1, s = 0, r[2] = {0, 0};
  double Jj[14], Jh[14], Jp[6], Jc[18];
  memset(Jc, 0, sizeof(Jc));
  const int ok = oracle_graph_obs_cam(kind, frames + 8 * (size_t)j, g->dof[j], frames + 8 * (size_t)h, g->dof[h], h == j, lm, lm_free,
                                  kind == 1 ? g->idp_anchor + 3 * (size_t)p : NULL,
                                  g->projection ? g->obs_bearing + 3 * (size_t)k : g->obs_xy + 2 * (size_t)k, info, g->huber, r,
                                  &w, &s, with_j ? Jj : NULL, with_j ? Jh : NULL, with_j ? Jp : NULL, g->projection, cam,
                                  g->intrinsics_free, with_j ? Jc : NULL);
  if (s_out) *s_out = s;
  if (!o) return ok;
  memset(o, 0, sizeof(*o));
  o->valid = ok;
  o->fj = j;
  o->fh = (kind == 1 && h != j) ? h : -1;
  o->lm = kind == 0 ? p : g->n_xyz + p;
  o->dp = lm_free ? (kind == 0 ? 3 : 1) : 0;
  if (!ok) return 0;
  o->r[0] = r[0]; o->r[1] = r[1]; o->w = w;
  o->L[0] = w; o->L[1] = 0; o->L[2] = 0; o->L[3] = w;
  if (info) for (int e = 0; e < 4; ++e) o->L[e] = w * info[e];
  if (with_j) {
    memcpy(o->Jj, Jj, sizeof(Jj));
    memcpy(o->Jh, Jh, sizeof(Jh));
    memcpy(o->Jp, Jp, sizeof(Jp));
    memcpy(o->Jc, Jc, sizeof(Jc));
  }
  return 1;
}

static double pose_edge_cost(const graph_problem* g, const double* S) {
  double cost = 0;
  for (int e = 0; e < g->n_edges; ++e) {
    double r[7];
    const int i = g->ei[e], j = g->ej[e];
    const int dim = oracle_pg_edge_residual(g->etype[e], S + 8 * (size_t)i, j >= 0 ? S + 8 * (size_t)j : S + 8 * (size_t)i,
                                            g->meas + 8 * (size_t)e, r);
    double q = 0;
    for (int a = 0; a < dim; ++a) {
      double Lr = 0;
      for (int b = 0; b < dim; ++b) Lr += (g->info ? g->info[49 * (size_t)e + 7 * a + b] : (a == b ? 1.0 : 0.0)) * r[b];
      q += r[a] * Lr;
    }
    cost += 0.5 * q;
  }
  return cost;
}

double oracle_graph_cost(const graph_problem* g) {
  double c = 0;
  for (int k = 0; k < g->n_obs; ++k) {
    double s;
    if (obs_eval(g, k, g->frames, g->xyz, g->idp_rho, g->intrinsics, NULL, 0, &s)) c += rho_huber(s, g->huber);
  }
  return 0.5 * c + pose_edge_cost(g, g->frames);
}

/* central-difference Jacobian of a pose edge (pg_oracle.c edge_jacobian, restated on the exported residual) */
static void pose_edge_jacobian(int type, const double* Si, const double* Sj, const double* meas, int which, int dof, double* J) {
  memset(J, 0, 49 * 8);
  for (int k = 0; k < 7; ++k) {
    if (!((dof >> k) & 1)) continue;
    double dp[7] = {0, 0, 0, 0, 0, 0, 0}, Sp[8], Sm[8], rp[7], rm[7];
    dp[k] = G_FD_STEP;
    oracle_sim3_retract(which == 0 ? Si : Sj, dp, Sp);
    dp[k] = -G_FD_STEP;
    oracle_sim3_retract(which == 0 ? Si : Sj, dp, Sm);
    const int dim = oracle_pg_edge_residual(type, which == 0 ? Sp : Si, which == 0 ? Sj : Sp, meas, rp);
    oracle_pg_edge_residual(type, which == 0 ? Sm : Si, which == 0 ? Sj : Sm, meas, rm);
    for (int a = 0; a < dim; ++a) J[7 * a + k] = (rp[a] - rm[a]) / (2.0 * G_FD_STEP);
  }
}

/* 3 x 3 SPD inverse (closed form); dims 1 and 3 */
static void inv_sym(const double* H, int dp, double* Hi) {
  memset(Hi, 0, 9 * 8);
  if (dp == 1) {
    Hi[0] = 1.0 / H[0];
    return;
  }
  const double a = H[0], b = H[1], c = H[2], d = H[4], e = H[5], f = H[8];
  const double A = d * f - e * e, B = c * e - b * f, C = b * e - c * d;
  const double det = a * A + b * B + c * C, id = 1.0 / det;
  Hi[0] = A * id; Hi[1] = B * id; Hi[2] = C * id;
  Hi[3] = B * id; Hi[4] = (a * f - c * c) * id; Hi[5] = (b * c - a * e) * id;
  Hi[6] = C * id; Hi[7] = Hi[5]; Hi[8] = (a * d - b * b) * id;
}

int oracle_graph_solve(graph_problem* g, const pg_options* opt, pg_summary* sum, int threads) {
  const int with_cam = g->intrinsics != NULL;
  if (with_cam && g->projection != 0) return 2;
  const int nf = g->n_frames, n = 7 * nf + (with_cam ? 9 : 0), nlm = g->n_xyz + g->n_idp, no = g->n_obs, ne = g->n_edges;
  double cam_new[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double* H = (double*)calloc((size_t)n * n, 8);   /* keyframe block of the normal equations, column-major */
  double* Hd = (double*)malloc((size_t)n * n * 8);
  double* gf = (double*)malloc((size_t)n * 8);
  double* d = (double*)malloc((size_t)n * 8);
  double* Hpp = (double*)calloc((size_t)(nlm ? nlm : 1) * 9, 8);
  double* gp = (double*)calloc((size_t)(nlm ? nlm : 1) * 3, 8);
  double* Hinv = (double*)calloc((size_t)(nlm ? nlm : 1) * 9, 8);
  double* dlm = (double*)calloc((size_t)(nlm ? nlm : 1) * 3, 8);
  obs_rec* rec = (obs_rec*)calloc((size_t)(no ? no : 1), sizeof(obs_rec));
  /* pose-edge records: A_ii, A_jj, A_ji (49 each), b_i, b_j (7 each) */
  double* erec = (double*)calloc((size_t)(ne ? ne : 1) * 161, 8);
  double* Snew = (double*)malloc((size_t)nf * 64);
  double* xyz_new = (double*)malloc((size_t)(g->n_xyz ? g->n_xyz : 1) * 24);
  double* rho_new = (double*)malloc((size_t)(g->n_idp ? g->n_idp : 1) * 8);
  /* observations grouped by landmark */
  int32_t* lstart = (int32_t*)calloc((size_t)nlm + 2, 4);
  int32_t* llist = (int32_t*)malloc((size_t)(no ? no : 1) * 4);
  for (int k = 0; k < no; ++k) lstart[(g->obs_kind[k] == 0 ? g->obs_point[k] : g->n_xyz + g->obs_point[k]) + 1]++;
  for (int p = 0; p < nlm; ++p) lstart[p + 1] += lstart[p];
  {
    int32_t* fill = (int32_t*)malloc((size_t)(nlm + 1) * 4);
    memcpy(fill, lstart, (size_t)(nlm + 1) * 4);
    for (int k = 0; k < no; ++k) llist[fill[g->obs_kind[k] == 0 ? g->obs_point[k] : g->n_xyz + g->obs_point[k]]++] = k;
    free(fill);
  }
  memset(sum, 0, sizeof(*sum));
  double radius = opt->initial_radius, decrease = 2.0;
  double cost = oracle_graph_cost(g);
  sum->initial_cost = cost;
  int need_lin = 1, term = 0, it = 0;
  for (it = 0; it < opt->max_iterations; ++it) {
    if (need_lin) {
      memset(H, 0, (size_t)n * n * 8);
      memset(gf, 0, (size_t)n * 8);
      memset(Hpp, 0, (size_t)(nlm ? nlm : 1) * 72);
      memset(gp, 0, (size_t)(nlm ? nlm : 1) * 24);
      /* pose-graph edges */
      for (int e = 0; e < ne; ++e) {
        const int i = g->ei[e], j = g->ej[e];
        const double* Si = g->frames + 8 * (size_t)i;
        const double* Sj = j >= 0 ? g->frames + 8 * (size_t)j : Si;
        double r[7], L[49], Ji[49], Jj[49], LJi[49], LJj[49], Lr[7];
        const int dim = oracle_pg_edge_residual(g->etype[e], Si, Sj, g->meas + 8 * (size_t)e, r);
        memset(L, 0, sizeof(L));
        for (int a = 0; a < dim; ++a)
          for (int b = 0; b < dim; ++b) L[7 * a + b] = g->info ? g->info[49 * (size_t)e + 7 * a + b] : (a == b ? 1.0 : 0.0);
        pose_edge_jacobian(g->etype[e], Si, Sj, g->meas + 8 * (size_t)e, 0, g->dof[i], Ji);
        if (j >= 0) pose_edge_jacobian(g->etype[e], Si, Sj, g->meas + 8 * (size_t)e, 1, g->dof[j], Jj);
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
        double* R = erec + 161 * (size_t)e;
        for (int p = 0; p < 7; ++p) {
          double gi = 0, gj = 0;
          for (int a = 0; a < dim; ++a) {
            gi += Ji[7 * a + p] * Lr[a];
            gj += Jj[7 * a + p] * Lr[a];
          }
          R[147 + p] = gi;
          R[154 + p] = gj;
          gf[7 * i + p] += gi;
          if (j >= 0) gf[7 * j + p] += gj;
          for (int q = 0; q < 7; ++q) {
            double hii = 0, hjj = 0, hji = 0;
            for (int a = 0; a < dim; ++a) {
              hii += Ji[7 * a + p] * LJi[7 * a + q];
              hjj += Jj[7 * a + p] * LJj[7 * a + q];
              hji += Jj[7 * a + p] * LJi[7 * a + q];
            }
            R[7 * p + q] = hii;
            R[49 + 7 * p + q] = hjj;
            R[98 + 7 * p + q] = hji;
            H[(size_t)(7 * i + q) * n + 7 * i + p] += hii;
            if (j >= 0) {
              H[(size_t)(7 * j + q) * n + 7 * j + p] += hjj;
              H[(size_t)(7 * i + q) * n + 7 * j + p] += hji;
              H[(size_t)(7 * j + p) * n + 7 * i + q] += hji;
            }
          }
        }
      }
      /* observations */
      for (int k = 0; k < no; ++k) {
        obs_rec* o = rec + k;
        if (!obs_eval(g, k, g->frames, g->xyz, g->idp_rho, g->intrinsics, o, 1, NULL)) continue;
        const double Lr[2] = {o->L[0] * o->r[0] + o->L[1] * o->r[1], o->L[2] * o->r[0] + o->L[3] * o->r[1]};
        const obs_slots T = slots_of(o, nf, with_cam);
        for (int x = 0; x < T.ns; ++x) {
          const int wx = T.width[x];
          for (int p = 0; p < wx; ++p) gf[T.base[x] + p] += T.J[x][p] * Lr[0] + T.J[x][wx + p] * Lr[1];
          for (int y = 0; y < T.ns; ++y) {
            const int wy = T.width[y];
            for (int p = 0; p < wx; ++p)
              for (int q = 0; q < wy; ++q) {
                const double LJ0 = o->L[0] * T.J[y][q] + o->L[1] * T.J[y][wy + q], LJ1 = o->L[2] * T.J[y][q] + o->L[3] * T.J[y][wy + q];
                H[(size_t)(T.base[y] + q) * n + T.base[x] + p] += T.J[x][p] * LJ0 + T.J[x][wx + p] * LJ1;
              }
          }
        }
        for (int a = 0; a < o->dp; ++a) {
          gp[3 * (size_t)o->lm + a] += o->Jp[a] * Lr[0] + o->Jp[3 + a] * Lr[1];
          for (int b = 0; b < o->dp; ++b) {
            const double LJ0 = o->L[0] * o->Jp[b] + o->L[1] * o->Jp[3 + b], LJ1 = o->L[2] * o->Jp[b] + o->L[3] * o->Jp[3 + b];
            Hpp[9 * (size_t)o->lm + 3 * a + b] += o->Jp[a] * LJ0 + o->Jp[3 + a] * LJ1;
          }
        }
      }
      double gmax = 0;
      for (int k = 0; k < n; ++k) gmax = fmax(gmax, fabs(gf[k]));
      for (int k = 0; k < 3 * nlm; ++k) gmax = fmax(gmax, fabs(gp[k]));
      if (gmax <= opt->gradient_tolerance) { term = 2; break; }
      need_lin = 0;
    }
    /* damped keyframe block, right-hand side, then the Schur complement of the landmarks */
    memcpy(Hd, H, (size_t)n * n * 8);
    for (int k = 0; k < n; ++k) {
      Hd[(size_t)k * n + k] += clampd(H[(size_t)k * n + k], 1e-6, 1e32) / radius;
      d[k] = -gf[k];
    }
    for (int p = 0; p < nlm; ++p) {
      int dp = 0;
      for (int q = lstart[p]; q < lstart[p + 1]; ++q)
        if (rec[llist[q]].valid && rec[llist[q]].dp > dp) dp = rec[llist[q]].dp;
      if (!dp) continue;
      double Hp[9];
      memcpy(Hp, Hpp + 9 * (size_t)p, 72);
      for (int a = 0; a < dp; ++a) Hp[4 * a] += clampd(Hp[4 * a], 1e-6, 1e32) / radius;
      double* Hi = Hinv + 9 * (size_t)p;
      inv_sym(Hp, dp, Hi);
      /* slots: every (observation, frame) pair of the landmark */
      for (int qa = lstart[p]; qa < lstart[p + 1]; ++qa) {
        const obs_rec* oa = rec + llist[qa];
        if (!oa->valid) continue;
        const obs_slots Ta = slots_of(oa, nf, with_cam);
        for (int x = 0; x < Ta.ns; ++x) {
          const int wa = Ta.width[x];
          double Wa[27], Ua[27];
          for (int r7 = 0; r7 < wa; ++r7)
            for (int b = 0; b < 3; ++b) {
              const double LJ0 = oa->L[0] * oa->Jp[b] + oa->L[1] * oa->Jp[3 + b], LJ1 = oa->L[2] * oa->Jp[b] + oa->L[3] * oa->Jp[3 + b];
              Wa[3 * r7 + b] = b < dp ? Ta.J[x][r7] * LJ0 + Ta.J[x][wa + r7] * LJ1 : 0.0;
            }
          for (int r7 = 0; r7 < wa; ++r7)
            for (int b = 0; b < 3; ++b) Ua[3 * r7 + b] = Wa[3 * r7] * Hi[b] + Wa[3 * r7 + 1] * Hi[3 + b] + Wa[3 * r7 + 2] * Hi[6 + b];
          for (int r7 = 0; r7 < wa; ++r7)
            d[Ta.base[x] + r7] += Ua[3 * r7] * gp[3 * (size_t)p] + Ua[3 * r7 + 1] * gp[3 * (size_t)p + 1] + Ua[3 * r7 + 2] * gp[3 * (size_t)p + 2];
          for (int qb = lstart[p]; qb < lstart[p + 1]; ++qb) {
            const obs_rec* ob = rec + llist[qb];
            if (!ob->valid) continue;
            const obs_slots Tb = slots_of(ob, nf, with_cam);
            for (int y = 0; y < Tb.ns; ++y) {
              const int wb = Tb.width[y];
              for (int c7 = 0; c7 < wb; ++c7) {
                double Wb[3];
                for (int b = 0; b < 3; ++b) {
                  const double LJ0 = ob->L[0] * ob->Jp[b] + ob->L[1] * ob->Jp[3 + b], LJ1 = ob->L[2] * ob->Jp[b] + ob->L[3] * ob->Jp[3 + b];
                  Wb[b] = b < dp ? Tb.J[y][c7] * LJ0 + Tb.J[y][wb + c7] * LJ1 : 0.0;
                }
                for (int r7 = 0; r7 < wa; ++r7)
                  Hd[(size_t)(Tb.base[y] + c7) * n + Ta.base[x] + r7] -= Ua[3 * r7] * Wb[0] + Ua[3 * r7 + 1] * Wb[1] + Ua[3 * r7 + 2] * Wb[2];
              }
            }
          }
        }
      }
    }
    int ok = oracle_potrf(Hd, n, threads) == 0;
    double new_cost = cost, model = 0, rho = -1;
    if (ok) {
      oracle_potrs(Hd, n, d);
      /* landmark steps: d_p = -Hinv (g_p + sum W^T d_f) */
      memset(dlm, 0, (size_t)(nlm ? nlm : 1) * 24);
      for (int p = 0; p < nlm; ++p) {
        int dp = 0;
        double t[3] = {gp[3 * (size_t)p], gp[3 * (size_t)p + 1], gp[3 * (size_t)p + 2]};
        for (int q = lstart[p]; q < lstart[p + 1]; ++q) {
          const obs_rec* o = rec + llist[q];
          if (!o->valid) continue;
          if (o->dp > dp) dp = o->dp;
          const obs_slots T = slots_of(o, nf, with_cam);
          for (int x = 0; x < T.ns; ++x) {
            const int wx = T.width[x];
            double Jd[2] = {0, 0};
            for (int k = 0; k < wx; ++k) {
              Jd[0] += T.J[x][k] * d[T.base[x] + k];
              Jd[1] += T.J[x][wx + k] * d[T.base[x] + k];
            }
            const double LJd[2] = {o->L[0] * Jd[0] + o->L[1] * Jd[1], o->L[2] * Jd[0] + o->L[3] * Jd[1]};
            for (int b = 0; b < o->dp; ++b) t[b] += o->Jp[b] * LJd[0] + o->Jp[3 + b] * LJd[1];
          }
        }
        if (!dp) continue;
        const double* Hi = Hinv + 9 * (size_t)p;
        for (int a = 0; a < dp; ++a) dlm[3 * (size_t)p + a] = -(Hi[3 * a] * t[0] + Hi[3 * a + 1] * t[1] + Hi[3 * a + 2] * t[2]);
      }
      /* model decrease, edge by edge, with the undamped Gauss-Newton Hessian: -( (J d)^T L r + 1/2 (J d)^T L (J d) ) */
      for (int e = 0; e < ne; ++e) {
        const double* R = erec + 161 * (size_t)e;
        const int i = g->ei[e], j = g->ej[e];
        const double* di = d + 7 * i;
        const double* dj = j >= 0 ? d + 7 * j : NULL;
        double lin = 0, quad = 0;
        for (int p = 0; p < 7; ++p) {
          lin += R[147 + p] * di[p] + (dj ? R[154 + p] * dj[p] : 0.0);
          for (int q = 0; q < 7; ++q) {
            quad += di[p] * R[7 * p + q] * di[q];
            if (dj) quad += dj[p] * R[49 + 7 * p + q] * dj[q] + 2.0 * dj[p] * R[98 + 7 * p + q] * di[q];
          }
        }
        model -= lin + 0.5 * quad;
      }
      for (int k = 0; k < no; ++k) {
        const obs_rec* o = rec + k;
        if (!o->valid) continue;
        double Jd[2] = {0, 0};
        for (int q = 0; q < 7; ++q) {
          Jd[0] += o->Jj[q] * d[7 * o->fj + q];
          Jd[1] += o->Jj[7 + q] * d[7 * o->fj + q];
          if (o->fh >= 0) {
            Jd[0] += o->Jh[q] * d[7 * o->fh + q];
            Jd[1] += o->Jh[7 + q] * d[7 * o->fh + q];
          }
        }
        if (with_cam)
          for (int q = 0; q < 9; ++q) {
            Jd[0] += o->Jc[q] * d[7 * nf + q];
            Jd[1] += o->Jc[9 + q] * d[7 * nf + q];
          }
        for (int b = 0; b < o->dp; ++b) {
          Jd[0] += o->Jp[b] * dlm[3 * (size_t)o->lm + b];
          Jd[1] += o->Jp[3 + b] * dlm[3 * (size_t)o->lm + b];
        }
        const double LJd[2] = {o->L[0] * Jd[0] + o->L[1] * Jd[1], o->L[2] * Jd[0] + o->L[3] * Jd[1]};
        const double Lr[2] = {o->L[0] * o->r[0] + o->L[1] * o->r[1], o->L[2] * o->r[0] + o->L[3] * o->r[1]};
        model -= (Jd[0] * Lr[0] + Jd[1] * Lr[1]) + 0.5 * (Jd[0] * LJd[0] + Jd[1] * LJd[1]);
      }
      /* candidate */
      for (int f = 0; f < nf; ++f) {
        if ((g->dof[f] & 127) == 0) memcpy(Snew + 8 * (size_t)f, g->frames + 8 * (size_t)f, 64);
        else oracle_sim3_retract(g->frames + 8 * (size_t)f, d + 7 * f, Snew + 8 * (size_t)f);
      }
      for (int p = 0; p < g->n_xyz; ++p)
        for (int a = 0; a < 3; ++a) xyz_new[3 * (size_t)p + a] = g->xyz[3 * (size_t)p + a] + dlm[3 * (size_t)p + a];
      for (int p = 0; p < g->n_idp; ++p) rho_new[p] = fmax(g->idp_rho[p] + dlm[3 * (size_t)(g->n_xyz + p)], 1e-9);
      if (with_cam)
        for (int q = 0; q < 9; ++q) cam_new[q] = g->intrinsics[q] + (((g->intrinsics_free >> q) & 1) ? d[7 * nf + q] : 0.0);
      double c = 0;
      for (int k = 0; k < no; ++k) {
        double s;
        if (obs_eval(g, k, Snew, xyz_new, rho_new, with_cam ? cam_new : NULL, NULL, 0, &s)) c += rho_huber(s, g->huber);
        else if (rec[k].valid) { c = INFINITY; break; }
      }
      new_cost = 0.5 * c + pose_edge_cost(g, Snew);
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
      memcpy(g->frames, Snew, (size_t)nf * 64);
      if (g->n_xyz) memcpy(g->xyz, xyz_new, (size_t)g->n_xyz * 24);
      if (g->n_idp) memcpy(g->idp_rho, rho_new, (size_t)g->n_idp * 8);
      if (with_cam) memcpy(g->intrinsics, cam_new, sizeof(cam_new));
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
  free(H); free(Hd); free(gf); free(d); free(Hpp); free(gp); free(Hinv); free(dlm); free(rec); free(erec); free(Snew);
  free(xyz_new); free(rho_new); free(lstart); free(llist);
  return term == 3 ? 1 : 0;
}
