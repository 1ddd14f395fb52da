// libgslam_optimizer.so — GSLAM Optimizer plugin backed by the MI355X HIP solver.
//
// Drop-in at GSLAM's own boundary: exports `createOptimizerInstance` through GSLAM_REGISTER_OPTIMIZER
// (GSLAM/core/Optimizer.h:42-51) so that GSLAM::Optimizer::create() (:234-248, default plugin name
// "libgslam_optimizer") loads it with no change to the host.  Implements
//   optimize(BundleGraph&)   (:229)      -> gh_ba_graph_* (mappoint bundle adjustment, SE3 keyframes); the graph stays in
//                                           HBM between calls and is rebuilt only when its topology changes
//                                        -> gh_pg_solve   when the graph holds se3Graph / sim3Graph / gpsGraph edges and no
//                                           point observations (pose-graph optimisation, SIM3 keyframes, UPDATE_KF_SCALE)
//   optimizePnP(...)         (:202-207)  -> gh_ba_pnp
//   optimizePose(...)        (:193-199)  -> gh_ba_pnp on the points of the first frame (anchor / idepth)
//   optimizeICP(...)         (:210-217)  -> gh_align_sim3 (3D-3D correspondences)
//   fitSim3(...)             (:220-225)  -> gh_align_sim3 (translations of two synchronised trajectories)
//                                        -> gh_graph_solve when the graph holds inverse-depth points (invDepths /
//                                           invDepthObserves, :106-111,152-153,160) or pose-graph edges TOGETHER with
//                                           point observations: the general solver (SIM3 keyframes, both landmark kinds)
//                                           and every graph with observations under PROJECTION_SPHERE
// Camera self-calibration (BundleGraph::camera + cameraDOF) of PinHole / OpenCV cameras goes through the general graph solve;
// still `return false` ("unsupported", as the interface allows): self-calibration of other camera models (CameraEstimationDOF names
// no flag for the ATAN model's parameter, and the reference's CameraATAN::Project disagrees with itself between its SSE and scalar
// paths off the z = 1 plane: Camera.h:291-294 vs :327-331) or under PROJECTION_SPHERE.
// (optimizePnP / optimizePose under PROJECTION_SPHERE go through the general graph solver: optimizePnPSphere.)
// Host code only; all arithmetic runs in libgslam_hip.so (no CPU fallback: no GPU => returns false).
#include <GSLAM/core/GSLAM.h>
#include <GSLAM/core/Optimizer.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>

#include "gslam_hip.h"

namespace {

class OptimizerHIP : public GSLAM::Optimizer {
 public:
  OptimizerHIP() : ctx_(nullptr), tried_(false) {}
  ~OptimizerHIP() override {
    if (graph_) gh_ba_graph_destroy(graph_);
    if (ctx_) gh_ctx_destroy(ctx_);
  }

  bool optimize(GSLAM::BundleGraph& graph) override {
    // Self-calibration (BundleGraph::camera + cameraDOF): the intrinsics join the unknowns of the general graph solve.
    if (graph.cameraDOF != GSLAM::UPDATE_CAMERA_NONE && graph.camera.isValid()) {
      if (_config.cameraProjectionType == GSLAM::PROJECTION_SPHERE) return unsupported("camera self-calibration under PROJECTION_SPHERE");
      const std::string type = graph.camera.CameraType();
      if (type != "PinHole" && type != "OpenCV") return unsupported("camera self-calibration of a camera that is neither PinHole nor OpenCV");
      if (calibrationMask(graph) != 0) return optimizeGeneral(graph);
    }
    const bool pose_edges = !graph.se3Graph.empty() || !graph.sim3Graph.empty() || !graph.gpsGraph.empty();
    const bool sphere = _config.cameraProjectionType == GSLAM::PROJECTION_SPHERE;
    const bool observations = !graph.mappointObserves.empty() || !graph.invDepthObserves.empty();
    if (!graph.invDepths.empty() || !graph.invDepthObserves.empty() || (pose_edges && !graph.mappointObserves.empty()) ||
        (sphere && observations))
      return optimizeGeneral(graph);  // the fast path below is pinhole bundle adjustment over XYZ points only
    if (pose_edges) return optimizePoseGraph(graph);
    if (graph.keyframes.empty()) return false;
    if (!context()) return false;

    BaArrays A;
    if (!to_problem(graph, A)) return false;
    const size_t nc = A.nc, np = A.np;
    std::vector<double>&pose = A.pose, &xyz = A.xyz, &oxy = A.oxy, &info = A.info;
    std::vector<int32_t>&dof = A.dof, &ocam = A.ocam, &opt = A.opt;
    std::vector<uint8_t>& pfree = A.pfree;
    const bool any_info = A.any_info;
    gh_ba_problem& pr = A.pr;
    gh_ba_options o;
    gh_ba_default_options(&o);
    o.huber_delta = _config.projectErrorHuberThreshold;
    o.max_iterations = _config.maxIterations;
    o.verbose = _config.verbose ? 1 : 0;
    gh_ba_summary s;
    // A back end optimises the same window again and again (local BA after every keyframe): the device-resident graph is
    // kept while the topology -- sizes, the (frame, point) pair of every observation, presence of information matrices --
    // is the one of the previous call; then only the values travel.  OptimizerHIP.CacheGraph=0 solves one-shot.
    gh_status st;
    {
      std::lock_guard<std::mutex> lock(mu_);
      if (svar.GetInt("OptimizerHIP.CacheGraph", 1) == 0) {
        st = gh_ba_solve(ctx_, &pr, &o, &s);
      } else {
        const bool same = graph_ != NULL && any_info == graph_info_ && ocam == graph_ocam_ && opt == graph_opt_ &&
                          nc == graph_nc_ && np == graph_np_;
        if (same) {
          st = gh_ba_graph_update(graph_, pose.data(), xyz.data(), oxy.data(), any_info ? info.data() : NULL, dof.data(),
                                  pfree.data());
          ++graph_hits_;
        } else {
          if (graph_) gh_ba_graph_destroy(graph_);
          graph_ = NULL;
          st = gh_ba_graph_create(ctx_, &pr, &o, &graph_);
          if (st == GH_OK) {
            graph_ocam_ = ocam; graph_opt_ = opt; graph_nc_ = nc; graph_np_ = np; graph_info_ = any_info;
          } else {
            graph_ = NULL;
          }
        }
        if (st == GH_OK) st = gh_ba_graph_solve(graph_, &o, &s);
        if (st == GH_OK) st = gh_ba_graph_read(graph_, pose.data(), xyz.data());
      }
    }
    if (st != GH_OK) {
      LOG(ERROR) << "OptimizerHIP: bundle adjustment failed (" << st << "): " << gh_last_error(ctx_);
      return false;
    }
    if (_config.verbose)
      LOG(INFO) << "OptimizerHIP: " << s.iterations << " LM iterations, cost " << s.initial_cost << " -> "
                << s.final_cost << " in " << s.total_ms << " ms";
    for (size_t i = 0; i < nc; ++i) {
      const double* p = &pose[i * 7];
      GSLAM::SIM3& T = graph.keyframes[i].estimation;
      T = GSLAM::SIM3(GSLAM::SO3(p[0], p[1], p[2], p[3]), GSLAM::Point3d(p[4], p[5], p[6]), T.get_scale());
    }
    for (size_t i = 0; i < np; ++i)
      graph.mappoints[i].first = GSLAM::Point3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    return true;
  }

  // A pinhole bundle graph over XYZ points flattened for gh_ba_problem (the vectors own the storage)
  struct BaArrays {
    size_t nc = 0, np = 0, no = 0;
    std::vector<double> pose, xyz, oxy, info;
    std::vector<int32_t> dof, ocam, opt;
    std::vector<uint8_t> pfree;
    bool any_info = false;
    gh_ba_problem pr;
  };
  bool to_problem(GSLAM::BundleGraph& graph, BaArrays& A) {
    const size_t nc = graph.keyframes.size(), np = graph.mappoints.size(), no = graph.mappointObserves.size();
    A.nc = nc; A.np = np; A.no = no;
    A.pose.assign(nc * 7, 0.0); A.xyz.assign(np * 3, 0.0); A.oxy.assign(no * 2, 0.0); A.info.clear();
    A.dof.assign(nc, 0); A.ocam.assign(no, 0); A.opt.assign(no, 0); A.pfree.assign(np, 0);
    std::vector<double>&pose = A.pose, &xyz = A.xyz, &oxy = A.oxy, &info = A.info;
    std::vector<int32_t>&dof = A.dof, &ocam = A.ocam, &opt = A.opt;
    std::vector<uint8_t>& pfree = A.pfree;
    for (size_t i = 0; i < nc; ++i) {
      const GSLAM::SIM3& T = graph.keyframes[i].estimation;  // T_wc, camera -> world
      const GSLAM::SO3 r = T.get_rotation();
      const GSLAM::Point3d t = T.get_translation();
      double* p = &pose[i * 7];
      p[0] = r.x; p[1] = r.y; p[2] = r.z; p[3] = r.w; p[4] = t.x; p[5] = t.y; p[6] = t.z;
      // SIM3 scale s > 0: X_c = T_wc^-1 X_w = R^T (X_w - t) / s (GSLAM/core/SIM3.h:120-131), and the pinhole residual
      // X_c.xy / X_c.z does not depend on s -- the SE3 part (R, t) is the whole problem, s is a gauge that mappoint
      // observations cannot see.  It is therefore neither optimised (UPDATE_KF_SCALE is ignored) nor changed: the
      // keyframe gets (R', t', s) back.  s <= 0 flips the depth sign and is rejected.
      if (!(T.get_scale() > 0)) return unsupported("keyframe with non-positive SIM3 scale");
      dof[i] = (int32_t)graph.keyframes[i].dof & GH_KF_SE3;
    }
    for (size_t i = 0; i < np; ++i) {
      xyz[3 * i] = graph.mappoints[i].first.x;
      xyz[3 * i + 1] = graph.mappoints[i].first.y;
      xyz[3 * i + 2] = graph.mappoints[i].first.z;
      pfree[i] = graph.mappoints[i].second ? 1 : 0;
    }
    bool any_info = false;
    for (size_t k = 0; k < no; ++k) any_info = any_info || graph.mappointObserves[k].information != NULL;
    A.any_info = any_info;
    if (any_info) info.resize(no * 4);
    for (size_t k = 0; k < no; ++k) {
      const GSLAM::BundleEdge& e = graph.mappointObserves[k];
      if (e.pointId >= np || e.frameId >= nc) {
        LOG(ERROR) << "OptimizerHIP: observation " << k << " references a missing vertex";
        return false;
      }
      opt[k] = (int32_t)e.pointId;
      ocam[k] = (int32_t)e.frameId;
      const double z = e.measurement.z;  // CameraAnchor: pinhole measurements live on the z = 1 plane (:58-61,102-103)
      if (!(z > 0)) {
        LOG(ERROR) << "OptimizerHIP: observation " << k << " has measurement.z = " << z << " (pinhole anchors need z > 0)";
        return false;
      }
      oxy[2 * k] = e.measurement.x / z;
      oxy[2 * k + 1] = e.measurement.y / z;
      if (any_info) {
        double* L = &info[4 * k];
        if (e.information) { L[0] = e.information[0]; L[1] = e.information[1]; L[2] = e.information[2]; L[3] = e.information[3]; }
        else { L[0] = 1; L[1] = 0; L[2] = 0; L[3] = 1; }
      }
    }
    gh_ba_problem& pr = A.pr;
    pr.n_cams = (int32_t)nc; pr.n_points = (int32_t)np; pr.n_obs = (int32_t)no;
    pr.cam_pose = pose.data(); pr.cam_dof = dof.data(); pr.point_xyz = xyz.data(); pr.point_free = pfree.data();
    pr.obs_cam = ocam.data(); pr.obs_point = opt.data(); pr.obs_xy = oxy.data();
    pr.obs_info = any_info ? info.data() : NULL;
    return true;
  }

  // "Convert bundle graph to pose graph" (GSLAM/core/Optimizer.h:230-232; the reference declares it and nothing else, so
  // the semantics are ours: gh_ba_marginalize / oracle_ba_marginalize).  Every pair of keyframes that shares at least
  // OptimizerHIP.MaginMinShared (default 15) map points becomes an SE3Edge {first < second, T_first^-1 T_second of the current
  // estimates, information = the two-view Schur complement over the shared points}; the map-point observations are
  // removed from the graph (they are what the edges stand for), keyframes and map points stay.  The information blocks the
  // edges point to are owned by this optimizer and live until its next magin() or its destruction.  Pinhole XYZ-point
  // graphs only (the fast path of optimize()).
  bool magin(GSLAM::BundleGraph& graph) override {
    if (_config.cameraProjectionType == GSLAM::PROJECTION_SPHERE) return unsupported("magin under PROJECTION_SPHERE");
    if (!graph.invDepths.empty() || !graph.invDepthObserves.empty()) return unsupported("magin of inverse-depth points");
    if (graph.keyframes.empty() || graph.mappointObserves.empty()) return false;
    if (!context()) return false;
    BaArrays A;
    if (!to_problem(graph, A)) return false;
    const int min_shared = svar.GetInt("OptimizerHIP.MaginMinShared", 15);
    std::lock_guard<std::mutex> lock(mu_);
    int32_t n = 0;
    gh_status st = gh_ba_marginalize(ctx_, &A.pr, _config.projectErrorHuberThreshold, min_shared, 0, NULL, NULL, NULL, NULL, &n);
    std::vector<int32_t> first((size_t)n), second((size_t)n);
    magin_info_.assign((size_t)n * 36, 0.0);
    if (st == GH_OK && n > 0)
      st = gh_ba_marginalize(ctx_, &A.pr, _config.projectErrorHuberThreshold, min_shared, n, first.data(), second.data(), NULL,
                             magin_info_.data(), &n);
    if (st != GH_OK) {
      LOG(ERROR) << "OptimizerHIP: magin failed (" << st << "): " << gh_last_error(ctx_);
      return false;
    }
    graph.se3Graph.reserve(graph.se3Graph.size() + (size_t)n);
    for (int32_t e = 0; e < n; ++e) {
      GSLAM::SE3Edge edge;
      edge.firstId = (GSLAM::FrameID)first[e];
      edge.secondId = (GSLAM::FrameID)second[e];
      const GSLAM::SIM3 &Ti = graph.keyframes[first[e]].estimation, &Tj = graph.keyframes[second[e]].estimation;
      const GSLAM::SE3 Ei(Ti.get_rotation(), Ti.get_translation()), Ej(Tj.get_rotation(), Tj.get_translation());
      edge.measurement = Ei.inverse() * Ej;
      edge.information = &magin_info_[(size_t)e * 36];
      graph.se3Graph.push_back(edge);
    }
    graph.mappointObserves.clear();
    if (_config.verbose) LOG(INFO) << "OptimizerHIP: magin: " << n << " SE3 edges from " << A.no << " observations";
    return true;
  }

  // the three pose-graph edge lists of a BundleGraph flattened for gh_pg_problem (the vectors own the storage)
  struct PoseEdges {
    std::vector<int32_t> f1, s1, f2, s2, fg;
    std::vector<double> m1, m2, mg, i1, i2, ig;
  };
  bool fill_pose_part(GSLAM::BundleGraph& graph, std::vector<double>& frames, std::vector<int32_t>& dof, PoseEdges& E,
                      gh_pg_problem& pr) {
    const size_t nf = graph.keyframes.size();
    bool any1 = false, any2 = false, anyg = false;
    for (size_t k = 0; k < graph.se3Graph.size(); ++k) any1 = any1 || graph.se3Graph[k].information != NULL;
    for (size_t k = 0; k < graph.sim3Graph.size(); ++k) any2 = any2 || graph.sim3Graph[k].information != NULL;
    for (size_t k = 0; k < graph.gpsGraph.size(); ++k) anyg = anyg || graph.gpsGraph[k].information != NULL;
    auto put_info = [](std::vector<double>& dst, const double* inf, int dim) {
      for (int a = 0; a < dim; ++a)
        for (int b = 0; b < dim; ++b) dst.push_back(inf ? inf[dim * a + b] : (a == b ? 1.0 : 0.0));
    };
    for (size_t k = 0; k < graph.se3Graph.size(); ++k) {
      const GSLAM::SE3Edge& e = graph.se3Graph[k];
      if (e.firstId >= nf || e.secondId >= nf || e.firstId == e.secondId) return bad_edge("se3Graph", k);
      E.f1.push_back((int32_t)e.firstId); E.s1.push_back((int32_t)e.secondId);
      double m[8];
      put_sim3(GSLAM::SIM3(e.measurement, 1.0), m);
      E.m1.insert(E.m1.end(), m, m + 7);
      if (any1) put_info(E.i1, e.information, 6);
    }
    for (size_t k = 0; k < graph.sim3Graph.size(); ++k) {
      const GSLAM::SIM3Edge& e = graph.sim3Graph[k];
      if (e.firstId >= nf || e.secondId >= nf || e.firstId == e.secondId || !(e.measurement.get_scale() > 0))
        return bad_edge("sim3Graph", k);
      E.f2.push_back((int32_t)e.firstId); E.s2.push_back((int32_t)e.secondId);
      double m[8];
      put_sim3(e.measurement, m);
      E.m2.insert(E.m2.end(), m, m + 8);
      if (any2) put_info(E.i2, e.information, 7);
    }
    for (size_t k = 0; k < graph.gpsGraph.size(); ++k) {
      const GSLAM::GPSEdge& e = graph.gpsGraph[k];
      if (e.frameId >= nf) return bad_edge("gpsGraph", k);
      E.fg.push_back((int32_t)e.frameId);
      double m[8];
      put_sim3(GSLAM::SIM3(e.measurement, 1.0), m);
      E.mg.insert(E.mg.end(), m, m + 7);
      if (anyg) put_info(E.ig, e.information, 6);
    }
    std::memset(&pr, 0, sizeof(pr));
    pr.n_frames = (int32_t)nf; pr.frame_sim3 = frames.data(); pr.frame_dof = dof.data();
    pr.n_se3 = (int32_t)E.f1.size(); pr.se3_first = E.f1.data(); pr.se3_second = E.s1.data(); pr.se3_meas = E.m1.data();
    pr.se3_info = any1 ? E.i1.data() : NULL;
    pr.n_sim3 = (int32_t)E.f2.size(); pr.sim3_first = E.f2.data(); pr.sim3_second = E.s2.data(); pr.sim3_meas = E.m2.data();
    pr.sim3_info = any2 ? E.i2.data() : NULL;
    pr.n_gps = (int32_t)E.fg.size(); pr.gps_frame = E.fg.data(); pr.gps_meas = E.mg.data(); pr.gps_info = anyg ? E.ig.data() : NULL;
    return true;
  }

  // Pose graph (loop closing / GPS fusion): keyframes are SIM3 T_wc, edges as Optimizer.h:127-148 defines them.
  bool optimizePoseGraph(GSLAM::BundleGraph& graph) {
    if (graph.keyframes.empty() || !context()) return false;
    const size_t nf = graph.keyframes.size();
    std::vector<double> frames(nf * 8);
    std::vector<int32_t> dof(nf);
    for (size_t i = 0; i < nf; ++i) {
      const GSLAM::SIM3& T = graph.keyframes[i].estimation;
      if (!(T.get_scale() > 0)) return unsupported("keyframe with non-positive SIM3 scale");
      put_sim3(T, &frames[i * 8]);
      dof[i] = (int32_t)graph.keyframes[i].dof & GH_KF_SIM3;
    }
    PoseEdges E;
    gh_pg_problem pr;
    if (!fill_pose_part(graph, frames, dof, E, pr)) return false;
    gh_ba_options o;
    gh_ba_default_options(&o);
    o.max_iterations = _config.maxIterations;
    o.verbose = _config.verbose ? 1 : 0;
    gh_ba_summary s;
    gh_status st;
    {
      std::lock_guard<std::mutex> lock(mu_);
      st = gh_pg_solve(ctx_, &pr, &o, &s);
    }
    if (st != GH_OK) {
      LOG(ERROR) << "OptimizerHIP: pose-graph optimisation failed (" << st << "): " << gh_last_error(ctx_);
      return false;
    }
    if (_config.verbose)
      LOG(INFO) << "OptimizerHIP: pose graph, " << s.iterations << " LM iterations, cost " << s.initial_cost << " -> "
                << s.final_cost << " in " << s.total_ms << " ms";
    for (size_t i = 0; i < nf; ++i) {
      const double* p = &frames[i * 8];
      graph.keyframes[i].estimation = GSLAM::SIM3(GSLAM::SO3(p[0], p[1], p[2], p[3]), GSLAM::Point3d(p[4], p[5], p[6]), p[7]);
    }
    return true;
  }

  // The general BundleGraph: SIM3 keyframes, pose-graph edges, XYZ map points and inverse-depth points together
  // (gh_graph_solve; specification in oracle/graph_oracle.c).  An inverse-depth point lives at anchor / idepth in the camera
  // of its host keyframe (InvDepthEstimation::frameId), the anchor taken on the z = 1 plane; UPDATE_ID_IDEPTH frees the
  // inverse depth, sigma is carried through untouched.  PROJECTION_SPHERE: anchors and measurements are bearings (normalised
  // to unit length here), the inverse depth is an inverse range, the residual lives in the tangent plane of the measurement.
  // CameraEstimationDOF (Optimizer.h:86-100) -> the free-parameter mask of gh_graph_problem::intrinsics (fx fy cx cy k1 k2 p1 p2 k3);
  // a PinHole camera has no distortion parameters to free
  static int calibrationMask(const GSLAM::BundleGraph& graph) {
    if (graph.cameraDOF == GSLAM::UPDATE_CAMERA_NONE || !graph.camera.isValid()) return 0;
    const int d = (int)graph.cameraDOF;
    int m = 0;
    if (d & GSLAM::UPDATE_CAMERA_FOCAL) m |= 3;
    if (d & GSLAM::UPDATE_CAMERA_CENTER) m |= 12;
    if (d & GSLAM::UPDATE_CAMERA_K1) m |= 1 << 4;
    if (d & GSLAM::UPDATE_CAMERA_K2) m |= 1 << 5;
    if (d & GSLAM::UPDATE_CAMERA_P1) m |= 1 << 6;
    if (d & GSLAM::UPDATE_CAMERA_P2) m |= 1 << 7;
    if (d & GSLAM::UPDATE_CAMERA_K3) m |= 1 << 8;
    if (graph.camera.CameraType() == "PinHole") m &= 15;
    return m;
  }

  // With a camera to calibrate (calibrationMask != 0): the measurements stay what they are everywhere else in GSLAM --
  // CameraAnchors, i.e. camera.UnProject(pixel) of the camera the graph carries -- and are taken back to pixels through that
  // camera's own Project; the solve then minimises the PIXEL error over keyframes, landmarks and the freed intrinsics
  // (gh_graph_problem::intrinsics), and graph.camera is replaced by the estimate.  projectErrorHuberThreshold and the 2x2
  // informations are given on the z = 1 plane: scaled by the focal lengths (threshold x sqrt(fx fy), information(a, b) /
  // (f_a f_b)).  Anchors of inverse-depth points keep the normalised value the initial camera gave them.
  bool optimizeGeneral(GSLAM::BundleGraph& graph) {
    const bool sphere = _config.cameraProjectionType == GSLAM::PROJECTION_SPHERE;
    const int calib = sphere ? 0 : calibrationMask(graph);
    double cam[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<double> cam_params;
    if (calib) {
      cam_params = graph.camera.getParameters();  // PinHole: w h fx fy cx cy; OpenCV: + k1 k2 p1 p2 k3
      if (cam_params.size() != 6 && cam_params.size() != 11) return unsupported("camera with an unexpected parameter list");
      for (size_t k = 2; k < cam_params.size(); ++k) cam[k - 2] = cam_params[k];
    }
    if (graph.keyframes.empty() || !context()) return false;
    const size_t nf = graph.keyframes.size(), np = graph.mappoints.size(), ni = graph.invDepths.size();
    std::vector<double> frames(nf * 8);
    std::vector<int32_t> dof(nf);
    for (size_t i = 0; i < nf; ++i) {
      const GSLAM::SIM3& T = graph.keyframes[i].estimation;
      if (!(T.get_scale() > 0)) return unsupported("keyframe with non-positive SIM3 scale");
      put_sim3(T, &frames[i * 8]);
      dof[i] = (int32_t)graph.keyframes[i].dof & GH_KF_SIM3;
    }
    PoseEdges E;
    gh_graph_problem gp;
    std::memset(&gp, 0, sizeof(gp));
    if (!fill_pose_part(graph, frames, dof, E, gp.pg)) return false;
    std::vector<double> xyz(np * 3), anchor(ni * 3), rho(ni), oxy, oinfo;
    std::vector<uint8_t> xfree(np), ifree(ni);
    std::vector<int32_t> host(ni), okind, opoint, oframe;
    for (size_t i = 0; i < np; ++i) {
      xyz[3 * i] = graph.mappoints[i].first.x;
      xyz[3 * i + 1] = graph.mappoints[i].first.y;
      xyz[3 * i + 2] = graph.mappoints[i].first.z;
      xfree[i] = graph.mappoints[i].second ? 1 : 0;
    }
    for (size_t i = 0; i < ni; ++i) {
      const GSLAM::InvDepthEstimation& v = graph.invDepths[i];
      const double an = sphere ? std::sqrt(v.anchor.x * v.anchor.x + v.anchor.y * v.anchor.y + v.anchor.z * v.anchor.z) : v.anchor.z;
      if (v.frameId >= nf || !(an > 0) || !(v.estimation.x > 0)) {
        LOG(ERROR) << "OptimizerHIP: inverse-depth point " << i << " needs a valid host keyframe, a usable anchor and idepth > 0";
        return false;
      }
      host[i] = (int32_t)v.frameId;
      anchor[3 * i] = v.anchor.x / an;
      anchor[3 * i + 1] = v.anchor.y / an;
      anchor[3 * i + 2] = v.anchor.z / an;
      rho[i] = v.estimation.x;
      ifree[i] = (v.dof & GSLAM::UPDATE_ID_IDEPTH) ? 1 : 0;
    }
    bool any_info = false;
    for (size_t k = 0; k < graph.mappointObserves.size(); ++k) any_info = any_info || graph.mappointObserves[k].information != NULL;
    for (size_t k = 0; k < graph.invDepthObserves.size(); ++k) any_info = any_info || graph.invDepthObserves[k].information != NULL;
    for (int kind = 0; kind < 2; ++kind) {
      const std::vector<GSLAM::BundleEdge>& obs = kind == 0 ? graph.mappointObserves : graph.invDepthObserves;
      for (size_t k = 0; k < obs.size(); ++k) {
        const GSLAM::BundleEdge& e = obs[k];
        const GSLAM::Point3d& m = e.measurement;
        const double mn = sphere ? std::sqrt(m.x * m.x + m.y * m.y + m.z * m.z) : m.z;
        if (e.pointId >= (kind == 0 ? np : ni) || e.frameId >= nf || !(mn > 0)) {
          LOG(ERROR) << "OptimizerHIP: " << (kind == 0 ? "mappoint" : "inverse-depth") << " observation " << k
                     << " references a missing vertex or has an unusable measurement";
          return false;
        }
        okind.push_back(kind);
        opoint.push_back((int32_t)e.pointId);
        oframe.push_back((int32_t)e.frameId);
        if (calib) {
          const GSLAM::Point2d px = graph.camera.Project(GSLAM::Point3d(m.x / mn, m.y / mn, 1.0));
          oxy.push_back(px.x);
          oxy.push_back(px.y);
        } else {
          oxy.push_back(m.x / mn);
          oxy.push_back(m.y / mn);
          if (sphere) oxy.push_back(m.z / mn);
        }
        if (any_info)
          for (int a = 0; a < 4; ++a) {
            const double v = e.information ? e.information[a] : ((a == 0 || a == 3) ? 1.0 : 0.0);
            oinfo.push_back(calib ? v / (cam[a >> 1] * cam[a & 1]) : v);
          }
      }
    }
    gp.n_xyz = (int32_t)np; gp.xyz = xyz.data(); gp.xyz_free = xfree.data();
    gp.n_idp = (int32_t)ni; gp.idp_host = host.data(); gp.idp_anchor = anchor.data(); gp.idp_rho = rho.data(); gp.idp_free = ifree.data();
    gp.n_obs = (int32_t)okind.size(); gp.obs_kind = okind.data(); gp.obs_point = opoint.data(); gp.obs_frame = oframe.data();
    gp.obs_info = any_info ? oinfo.data() : NULL;
    gp.projection = sphere ? 1 : 0;
    (sphere ? gp.obs_bearing : gp.obs_xy) = oxy.data();
    if (calib) {
      gp.intrinsics = cam;
      gp.intrinsics_free = calib;
    }
    gh_ba_options o;
    gh_ba_default_options(&o);
    o.huber_delta = _config.projectErrorHuberThreshold * (calib ? std::sqrt(std::fabs(cam[0] * cam[1])) : 1.0);
    o.max_iterations = _config.maxIterations;
    o.verbose = _config.verbose ? 1 : 0;
    gh_ba_summary s;
    gh_status st;
    {
      std::lock_guard<std::mutex> lock(mu_);
      st = gh_graph_solve(ctx_, &gp, &o, &s);
    }
    if (st != GH_OK) {
      LOG(ERROR) << "OptimizerHIP: graph optimisation failed (" << st << "): " << gh_last_error(ctx_);
      return false;
    }
    if (_config.verbose)
      LOG(INFO) << "OptimizerHIP: general graph, " << s.iterations << " LM iterations, cost " << s.initial_cost << " -> "
                << s.final_cost << " in " << s.total_ms << " ms";
    for (size_t i = 0; i < nf; ++i) {
      const double* p = &frames[i * 8];
      graph.keyframes[i].estimation = GSLAM::SIM3(GSLAM::SO3(p[0], p[1], p[2], p[3]), GSLAM::Point3d(p[4], p[5], p[6]), p[7]);
    }
    for (size_t i = 0; i < np; ++i) graph.mappoints[i].first = GSLAM::Point3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    for (size_t i = 0; i < ni; ++i) graph.invDepths[i].estimation.x = rho[i];
    if (calib) {
      for (size_t k = 2; k < cam_params.size(); ++k) cam_params[k] = cam[k - 2];
      graph.camera = GSLAM::Camera(cam_params);
      if (_config.verbose) LOG(INFO) << "OptimizerHIP: calibrated camera " << graph.camera.info();
    }
    return true;
  }

  // TRACKING with known depth in the first frame (Optimizer.h:193-199): the point of match k lives at anchor1 / idepth in
  // frame 1, its projection into frame 2 is matched against anchor2; relativePose = T_12 (P_1 = T_12 P_2, :131-134) is the
  // camera-to-"world" pose of camera 2 with frame 1 as the world, i.e. a PnP problem.  Matches whose inverse depth is
  // not positive (unknown) are left out; the depth estimates themselves are not updated.
  bool optimizePose(std::vector<std::pair<GSLAM::CameraAnchor, GSLAM::CameraAnchor> >& matches,
                    std::vector<GSLAM::IdepthEstimation>& firstIDepth, GSLAM::SE3& relativePose,
                    GSLAM::KeyFrameEstimzationDOF dof = GSLAM::UPDATE_KF_SE3, double* information = NULL) override {
    if (matches.size() != firstIDepth.size() || !context()) return false;
    const bool sphere = _config.cameraProjectionType == GSLAM::PROJECTION_SPHERE;
    std::vector<std::pair<GSLAM::Point3d, GSLAM::CameraAnchor> > m3d;
    for (size_t k = 0; k < matches.size(); ++k) {
      const GSLAM::CameraAnchor& a = matches[k].first;
      const double rho = firstIDepth[k].x;
      if (sphere) {  // the anchor is a bearing, the inverse depth an inverse RANGE (Optimizer.h:58-61,102-103)
        const double an = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
        if (!(rho > 0) || !(an > 0)) continue;
        const double d = 1.0 / (rho * an);
        m3d.push_back(std::make_pair(GSLAM::Point3d(a.x * d, a.y * d, a.z * d), matches[k].second));
        continue;
      }
      const double z1 = a.z;
      if (!(rho > 0) || !(z1 > 0)) continue;
      const double d = 1.0 / (rho * z1);  // the anchor is on the z = 1 plane after division by its z
      m3d.push_back(std::make_pair(GSLAM::Point3d(a.x * d, a.y * d, 1.0 / rho), matches[k].second));
    }
    if (m3d.size() < 3) return false;
    return optimizePnP(m3d, relativePose, dof, information);
  }

  // 3D-3D correspondences (first, second): pose maps the FIRST point set onto the SECOND, second ~ pose * first, in closed
  // form (Horn).  dof & UPDATE_KF_SCALE decides whether the scale is estimated.  information: 7 x 7 row-major.
  bool optimizeICP(const std::vector<std::pair<GSLAM::Point3d, GSLAM::Point3d> >& matches, GSLAM::SIM3& pose,
                   GSLAM::KeyFrameEstimzationDOF dof = GSLAM::UPDATE_KF_SE3, double* information = NULL) override {
    std::vector<double> a, b;
    for (size_t k = 0; k < matches.size(); ++k) {
      a.push_back(matches[k].first.x); a.push_back(matches[k].first.y); a.push_back(matches[k].first.z);
      b.push_back(matches[k].second.x); b.push_back(matches[k].second.y); b.push_back(matches[k].second.z);
    }
    return align(a, b, pose, dof, information);
  }

  // Two synchronised trajectories (pairs of poses of the same instants): the similarity that maps the translations of the
  // first trajectory onto those of the second (what an evaluation / map-merging step needs; scale with UPDATE_KF_SIM3).
  bool fitSim3(const std::vector<std::pair<GSLAM::SE3, GSLAM::SE3> >& matches, GSLAM::SIM3& sim3,
               GSLAM::KeyFrameEstimzationDOF dof = GSLAM::UPDATE_KF_SIM3, double* information = NULL) override {
    std::vector<double> a, b;
    for (size_t k = 0; k < matches.size(); ++k) {
      const GSLAM::Point3d p = matches[k].first.get_translation(), q = matches[k].second.get_translation();
      a.push_back(p.x); a.push_back(p.y); a.push_back(p.z);
      b.push_back(q.x); b.push_back(q.y); b.push_back(q.z);
    }
    return align(a, b, sim3, dof, information);
  }

  bool optimizePnP(const std::vector<std::pair<GSLAM::Point3d, GSLAM::CameraAnchor> >& matches, GSLAM::SE3& pose,
                   GSLAM::KeyFrameEstimzationDOF dof = GSLAM::UPDATE_KF_SE3, double* information = NULL) override {
    if (matches.empty() || !context()) return false;
    if (_config.cameraProjectionType == GSLAM::PROJECTION_SPHERE) return optimizePnPSphere(matches, pose, dof, information);
    const size_t n = matches.size();
    std::vector<double> X(n * 3), m(n * 2);
    for (size_t k = 0; k < n; ++k) {
      X[3 * k] = matches[k].first.x; X[3 * k + 1] = matches[k].first.y; X[3 * k + 2] = matches[k].first.z;
      const double z = matches[k].second.z;
      if (!(z > 0)) {
        LOG(ERROR) << "OptimizerHIP: match " << k << " has anchor z = " << z << " (pinhole anchors need z > 0)";
        return false;
      }
      m[2 * k] = matches[k].second.x / z;
      m[2 * k + 1] = matches[k].second.y / z;
    }
    const GSLAM::SO3& r = pose.get_rotation();
    const GSLAM::Point3d& t = pose.get_translation();
    double p[7] = {r.x, r.y, r.z, r.w, t.x, t.y, t.z};
    gh_ba_options o;
    gh_ba_default_options(&o);
    o.huber_delta = _config.projectErrorHuberThreshold;
    o.max_iterations = _config.maxIterations;
    gh_ba_summary s;
    const gh_status st = gh_ba_pnp(ctx_, X.data(), m.data(), (int)n, p, (int)dof & GH_KF_SE3, &o, information, &s);
    if (st != GH_OK) {
      LOG(ERROR) << "OptimizerHIP: gh_ba_pnp failed (" << st << "): " << gh_last_error(ctx_);
      return false;
    }
    pose = GSLAM::SE3(GSLAM::SO3(p[0], p[1], p[2], p[3]), GSLAM::Point3d(p[4], p[5], p[6]));
    return true;
  }

 private:
  // optimizePnP under PROJECTION_SPHERE (Optimizer.h:58-61,174-176,202-207): the measurements are bearings, the residual of
  // a match is the predicted bearing in the tangent plane of the measured one.  That is the observation model of the general
  // graph solver (gh_graph_solve, projection = 1), so the problem goes there as ONE free keyframe observing n fixed map
  // points.  information (6 x 6, row-major, [t r] order of the dof bits): sum of J^T J of the tangent-plane residuals at the
  // solution w.r.t. the right-multiplicative update T_wc <- T_wc exp(delta), by central differences on the host (n x 12
  // evaluations of a 3-vector), rows / columns of masked dof zero.
  bool optimizePnPSphere(const std::vector<std::pair<GSLAM::Point3d, GSLAM::CameraAnchor> >& matches, GSLAM::SE3& pose,
                         GSLAM::KeyFrameEstimzationDOF dof, double* information) {
    const size_t n = matches.size();
    std::vector<double> xyz(n * 3), bearing(n * 3);
    std::vector<uint8_t> xfree(n, 0);
    std::vector<int32_t> okind(n, 0), opoint(n), oframe(n, 0);
    for (size_t k = 0; k < n; ++k) {
      const GSLAM::Point3d& m = matches[k].second;
      const double mn = std::sqrt(m.x * m.x + m.y * m.y + m.z * m.z);
      if (!(mn > 0)) {
        LOG(ERROR) << "OptimizerHIP: match " << k << " has a zero bearing";
        return false;
      }
      xyz[3 * k] = matches[k].first.x; xyz[3 * k + 1] = matches[k].first.y; xyz[3 * k + 2] = matches[k].first.z;
      bearing[3 * k] = m.x / mn; bearing[3 * k + 1] = m.y / mn; bearing[3 * k + 2] = m.z / mn;
      opoint[k] = (int32_t)k;
    }
    double frame[8];
    put_sim3(GSLAM::SIM3(pose, 1.0), frame);
    int32_t fdof = (int32_t)dof & GH_KF_SE3;
    gh_graph_problem gp;
    std::memset(&gp, 0, sizeof(gp));
    gp.pg.n_frames = 1;
    gp.pg.frame_sim3 = frame;
    gp.pg.frame_dof = &fdof;
    gp.n_xyz = (int32_t)n; gp.xyz = xyz.data(); gp.xyz_free = xfree.data();
    gp.n_obs = (int32_t)n; gp.obs_kind = okind.data(); gp.obs_point = opoint.data(); gp.obs_frame = oframe.data();
    gp.projection = 1;
    gp.obs_bearing = bearing.data();
    gh_ba_options o;
    gh_ba_default_options(&o);
    o.huber_delta = _config.projectErrorHuberThreshold;
    o.max_iterations = _config.maxIterations;
    gh_ba_summary s;
    gh_status st;
    {
      std::lock_guard<std::mutex> lock(mu_);
      st = gh_graph_solve(ctx_, &gp, &o, &s);
    }
    if (st != GH_OK && st != GH_ERR_NUMERIC) {  // (GH_ERR_NUMERIC: the start was the optimum already, the pose is valid)
      LOG(ERROR) << "OptimizerHIP: sphere PnP failed (" << st << "): " << gh_last_error(ctx_);
      return false;
    }
    pose = GSLAM::SE3(GSLAM::SO3(frame[0], frame[1], frame[2], frame[3]), GSLAM::Point3d(frame[4], frame[5], frame[6]));
    if (information) {
      for (int i = 0; i < 36; ++i) information[i] = 0.0;
      const double h = 1e-6;
      for (size_t k = 0; k < n; ++k) {
        const GSLAM::Point3d X = matches[k].first, b(bearing[3 * k], bearing[3 * k + 1], bearing[3 * k + 2]);
        // an orthonormal basis of the tangent plane at b (J^T J does not depend on which)
        GSLAM::Point3d e1 = std::fabs(b.x) < 0.9 ? GSLAM::Point3d(1, 0, 0) : GSLAM::Point3d(0, 1, 0);
        e1 = e1 - b * (e1.x * b.x + e1.y * b.y + e1.z * b.z);
        e1 = e1 / std::sqrt(e1.x * e1.x + e1.y * e1.y + e1.z * e1.z);
        const GSLAM::Point3d e2(b.y * e1.z - b.z * e1.y, b.z * e1.x - b.x * e1.z, b.x * e1.y - b.y * e1.x);
        double J[2][6];
        for (int c = 0; c < 6; ++c) {
          double r[2][2];
          for (int sgn = 0; sgn < 2; ++sgn) {
            // exp(delta) for a delta along one axis: a pure translation (c < 3) or a pure rotation (the reference's SE3::exp
            // divides by the rotation angle, which is zero for the former)
            GSLAM::Point3d axis(0, 0, 0);
            (c % 3 == 0 ? axis.x : (c % 3 == 1 ? axis.y : axis.z)) = sgn == 0 ? h : -h;
            const GSLAM::SE3 step = c < 3 ? GSLAM::SE3(GSLAM::SO3(), axis) : GSLAM::SE3(GSLAM::SO3::exp(axis), GSLAM::Point3d(0, 0, 0));
            const GSLAM::Point3d Xc = (pose * step).inverse() * X;
            const double xn = std::sqrt(Xc.x * Xc.x + Xc.y * Xc.y + Xc.z * Xc.z);
            const GSLAM::Point3d u = Xc / xn;
            r[sgn][0] = e1.x * u.x + e1.y * u.y + e1.z * u.z;
            r[sgn][1] = e2.x * u.x + e2.y * u.y + e2.z * u.z;
          }
          J[0][c] = (((int)dof >> c) & 1) ? (r[0][0] - r[1][0]) / (2 * h) : 0.0;
          J[1][c] = (((int)dof >> c) & 1) ? (r[0][1] - r[1][1]) / (2 * h) : 0.0;
        }
        for (int a = 0; a < 6; ++a)
          for (int c = 0; c < 6; ++c) information[6 * a + c] += J[0][a] * J[0][c] + J[1][a] * J[1][c];
      }
    }
    return true;
  }

  static void put_sim3(const GSLAM::SIM3& T, double* p) {
    const GSLAM::SO3 r = T.get_rotation();
    const GSLAM::Point3d t = T.get_translation();
    p[0] = r.x; p[1] = r.y; p[2] = r.z; p[3] = r.w; p[4] = t.x; p[5] = t.y; p[6] = t.z; p[7] = T.get_scale();
  }
  bool bad_edge(const char* list, size_t k) {
    LOG(ERROR) << "OptimizerHIP: " << list << "[" << k << "] references a missing keyframe, connects a keyframe with itself or "
               << "carries a non-positive scale";
    return false;
  }
  bool align(const std::vector<double>& a, const std::vector<double>& b, GSLAM::SIM3& out, int dof, double* information) {
    if (a.size() < 9 || !context()) return false;
    double S[8], ssq = 0;
    int ok = 0;
    gh_status st;
    {
      std::lock_guard<std::mutex> lock(mu_);
      st = gh_align_sim3(ctx_, a.data(), b.data(), (int)(a.size() / 3), dof & GH_KF_SIM3, S, information, &ssq, &ok);
    }
    if (st != GH_OK) {
      LOG(ERROR) << "OptimizerHIP: gh_align_sim3 failed (" << st << "): " << gh_last_error(ctx_);
      return false;
    }
    if (!ok) return false;  // degenerate correspondences
    out = GSLAM::SIM3(GSLAM::SO3(S[0], S[1], S[2], S[3]), GSLAM::Point3d(S[4], S[5], S[6]), S[7]);
    return true;
  }
  bool unsupported(const char* what) {
    LOG(WARNING) << "OptimizerHIP: " << what << " is not supported by the HIP optimizer";
    return false;
  }
  // The context is created lazily on the calling thread: GSLAM constructs optimizers on one thread and
  // calls them from Messenger workers (GSLAM/core/Messenger.h:248).
  bool context() {
    std::lock_guard<std::mutex> lock(mu_);
    if (!ctx_ && !tried_) {
      tried_ = true;
      const int dev = svar.GetInt("OptimizerHIP.Device", 0);
      if (gh_ctx_create(dev, &ctx_) != GH_OK) {
        ctx_ = nullptr;
        LOG(ERROR) << "OptimizerHIP: no usable HIP device " << dev << " (there is no CPU fallback)";
      } else {
        // linear solver of the reduced camera system: "auto" (band solver on trajectory graphs, else dense), "dense", "band"
        const std::string solver = svar.GetString("OptimizerHIP.Solver", "auto");
        gh_ctx_set_ba_solver(ctx_, solver == "dense" ? GH_BA_SOLVER_DENSE : (solver == "band" ? GH_BA_SOLVER_BAND : GH_BA_SOLVER_AUTO));
      }
    }
    return ctx_ != nullptr;
  }
  gh_ctx* ctx_;
  bool tried_;
  std::mutex mu_;
  // device-resident graph of the previous optimize() and the topology it was built for
  gh_ba_graph* graph_ = NULL;
  std::vector<int32_t> graph_ocam_, graph_opt_;
  size_t graph_nc_ = 0, graph_np_ = 0, graph_hits_ = 0;
  bool graph_info_ = false;
  std::vector<double> magin_info_;  // 6 x 6 information blocks the edges of the last magin() point to
};

}  // namespace

GSLAM_REGISTER_OPTIMIZER(OptimizerHIP);
