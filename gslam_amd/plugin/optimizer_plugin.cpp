// libgslam_optimizer.so — GSLAM Optimizer plugin backed by the MI355X HIP solver.
//
// Drop-in at GSLAM's own boundary: exports `createOptimizerInstance` through GSLAM_REGISTER_OPTIMIZER
// (GSLAM/core/Optimizer.h:42-51) so that GSLAM::Optimizer::create() (:234-248, default plugin name
// "libgslam_optimizer") loads it with no change to the host.  Implements
//   optimize(BundleGraph&)   (:229)      -> gh_ba_graph_* (mappoint bundle adjustment, SE3 keyframes); the graph stays in
//                                           HBM between calls and is rebuilt only when its topology changes
//   optimizePnP(...)         (:202-207)  -> gh_ba_pnp
// Everything else keeps the base-class default `return false` ("unsupported"), as the interface allows.
// Host code only; all arithmetic runs in libgslam_hip.so (no CPU fallback: no GPU => returns false).
#include <GSLAM/core/GSLAM.h>
#include <GSLAM/core/Optimizer.h>

#include <cstdint>
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
    // supported sub-problem: xyz map points observed by SE3/SIM3 keyframes, pinhole anchors
    if (_config.cameraProjectionType != GSLAM::PROJECTION_PINHOLE) return unsupported("sphere projection");
    if (!graph.invDepths.empty() || !graph.invDepthObserves.empty()) return unsupported("inverse-depth points");
    if (!graph.se3Graph.empty() || !graph.sim3Graph.empty() || !graph.gpsGraph.empty())
      return unsupported("pose-graph / GPS edges");
    if (graph.cameraDOF != GSLAM::UPDATE_CAMERA_NONE && graph.camera.isValid())
      return unsupported("camera self-calibration");
    if (graph.keyframes.empty()) return false;
    if (!context()) return false;

    const size_t nc = graph.keyframes.size(), np = graph.mappoints.size(), no = graph.mappointObserves.size();
    std::vector<double> pose(nc * 7), xyz(np * 3), oxy(no * 2), info;
    std::vector<int32_t> dof(nc), ocam(no), opt(no);
    std::vector<uint8_t> pfree(np);
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
    gh_ba_problem pr;
    pr.n_cams = (int32_t)nc; pr.n_points = (int32_t)np; pr.n_obs = (int32_t)no;
    pr.cam_pose = pose.data(); pr.cam_dof = dof.data(); pr.point_xyz = xyz.data(); pr.point_free = pfree.data();
    pr.obs_cam = ocam.data(); pr.obs_point = opt.data(); pr.obs_xy = oxy.data();
    pr.obs_info = any_info ? info.data() : NULL;
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

  bool optimizePnP(const std::vector<std::pair<GSLAM::Point3d, GSLAM::CameraAnchor> >& matches, GSLAM::SE3& pose,
                   GSLAM::KeyFrameEstimzationDOF dof = GSLAM::UPDATE_KF_SE3, double* information = NULL) override {
    if (_config.cameraProjectionType != GSLAM::PROJECTION_PINHOLE) return unsupported("sphere projection");
    if (matches.empty() || !context()) return false;
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
};

}  // namespace

GSLAM_REGISTER_OPTIMIZER(OptimizerHIP);
