// libgslam_estimator.so — GSLAM::Estimator plugin (GSLAM/core/Estimator.h:92-191) on the MI355X.
// Exports `createEstimatorInstance` through the reference's own USE_ESTIMATOR_PLUGIN macro (:42-53), so
// GSLAM::Estimator::create() (default plugin name "libgslam_estimator", svar EstimatorPlugin) loads it unchanged.
// All nine pure virtuals are implemented: findHomography / findAffine2D / findFundamental / findEssentialMatrix / findSIM3 /
// findAffine3D / findPlane / findPnP -> gh_ransac_estimate_ex (inlier masks, `confidence` honoured), trianglate -> gh_triangulate.
// Sampling (`method & SAMPLE_METHOD`, Estimator.h:86-89):
//   RANSAC    the default.
//   NOSAMPLE  (2 << 8) the least-squares model of all correspondences, mask by `threshold`.
//   LMEDS     (1 << 8) least median of squares.  In the reference's enum the model ids follow MODEL_METHOD = 0xFF, i.e. they are
//             0x100..0x111 and collide with this bit (F8_Point == LMEDS == 0x100; the interface's own defaults are
//             `X & RANSAC` = 0), so a caller cannot flag it next to a model id.  It is taken when `method` is exactly LMEDS in a
//             function whose model is not F8_Point (every one but findFundamental), or for every call with svar
//             `EstimatorHIP.Sampling` = "LMEDS" ("NOSAMPLE" / "RANSAC" force those).
#include <GSLAM/core/GSLAM.h>
#include <GSLAM/core/Estimator.h>

#include <mutex>
#include <vector>

#include "gslam_hip.h"

namespace {

class EstimatorHIP : public GSLAM::Estimator {
 public:
  EstimatorHIP() : ctx_(nullptr) {}
  ~EstimatorHIP() override {
    if (ctx_) gh_ctx_destroy(ctx_);
  }
  std::string type() const override { return "EstimatorHIP"; }

  bool findHomography(GSLAM::Homography2D* H, const std::vector<GSLAM::Point2d>& src,
                      const std::vector<GSLAM::Point2d>& dst, int method, double threshold, double confidence,
                      std::vector<uchar>* mask) const override {
    double m[12];
    if (!run(GH_MODEL_HOMOGRAPHY, src, dst, method, threshold, confidence, m, mask)) return false;
    if (H) for (int i = 0; i < 9; ++i) H->data()[i] = m[i];
    return true;
  }
  bool findAffine2D(GSLAM::Affine2D* A, const std::vector<GSLAM::Point2d>& src, const std::vector<GSLAM::Point2d>& dst,
                    int method, double threshold, double confidence, std::vector<uchar>* mask) const override {
    double m[12];
    if (!run(GH_MODEL_AFFINE2D, src, dst, method, threshold, confidence, m, mask)) return false;
    if (A) for (int i = 0; i < 6; ++i) A->data()[i] = m[i];
    return true;
  }
  bool findFundamental(GSLAM::Fundamental* F, const std::vector<GSLAM::Point2d>& p1,
                       const std::vector<GSLAM::Point2d>& p2, int method, double threshold, double confidence,
                       std::vector<uchar>* mask) const override {
    double m[12];
    if (!run(GH_MODEL_FUNDAMENTAL, p1, p2, method, threshold, confidence, m, mask)) return false;
    if (F) for (int i = 0; i < 9; ++i) F->data()[i] = m[i];
    return true;
  }
  bool findAffine3D(GSLAM::Affine3D* A, const std::vector<GSLAM::Point3d>& src, const std::vector<GSLAM::Point3d>& dst,
                    int method, double threshold, double confidence, std::vector<uchar>* mask) const override {
    if (src.size() != dst.size() || !context()) return false;
    double m[12];
    if (!estimate(GH_MODEL_AFFINE3D, (const double*)src.data(), (const double*)dst.data(), (int)src.size(), threshold, confidence, m,
                  mask, method))
      return false;
    if (A) for (int i = 0; i < 12; ++i) A->data()[i] = m[i];
    return true;
  }
  // points1 / points2 are normalised image coordinates (camera.UnProject), threshold in the same units
  bool findEssentialMatrix(GSLAM::Essential* E, const std::vector<GSLAM::Point2d>& p1, const std::vector<GSLAM::Point2d>& p2,
                           int method, double threshold, double confidence, std::vector<uchar>* mask) const override {
    double m[12];
    if (!run(GH_MODEL_ESSENTIAL, p1, p2, method, threshold, confidence, m, mask)) return false;
    if (E) for (int i = 0; i < 9; ++i) E->data()[i] = m[i];
    return true;
  }
  // to ~ s R from + t (Horn's closed form inside RANSAC)
  bool findSIM3(GSLAM::SIM3* S, const std::vector<GSLAM::Point3d>& from, const std::vector<GSLAM::Point3d>& to, int method,
                double threshold, double confidence, std::vector<uchar>* mask) const override {
    if (from.size() != to.size() || !context()) return false;
    double m[12];
    if (!estimate(GH_MODEL_SIM3, (const double*)from.data(), (const double*)to.data(), (int)from.size(), threshold, confidence, m, mask,
                  method))
      return false;
    if (S) *S = GSLAM::SIM3(GSLAM::SO3(m[0], m[1], m[2], m[3]), GSLAM::Point3d(m[4], m[5], m[6]), m[7]);
    return true;
  }
  // plane pose: origin = the plane point closest to the world origin, z axis = the unit normal
  bool findPlane(GSLAM::SE3* plane, const std::vector<GSLAM::Point3d>& points, int method, double threshold,
                 double confidence, std::vector<uchar>* mask) const override {
    if (!context()) return false;
    double m[12];
    if (!estimate(GH_MODEL_PLANE, (const double*)points.data(), (const double*)points.data(), (int)points.size(), threshold, confidence,
                  m, mask, method))
      return false;
    if (plane) {
      const double n[3] = {m[0], m[1], m[2]};
      const double ref[3] = {fabs(n[0]) < 0.9 ? 1.0 : 0.0, fabs(n[0]) < 0.9 ? 0.0 : 1.0, 0.0};
      double x[3] = {ref[1] * n[2] - ref[2] * n[1], ref[2] * n[0] - ref[0] * n[2], ref[0] * n[1] - ref[1] * n[0]};
      const double xn = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
      for (int e = 0; e < 3; ++e) x[e] /= xn;
      const double y[3] = {n[1] * x[2] - n[2] * x[1], n[2] * x[0] - n[0] * x[2], n[0] * x[1] - n[1] * x[0]};
      double R[9] = {x[0], y[0], n[0], x[1], y[1], n[1], x[2], y[2], n[2]};
      GSLAM::SO3 rot;
      rot.fromMatrix(R);
      *plane = GSLAM::SE3(rot, GSLAM::Point3d(-m[3] * n[0], -m[3] * n[1], -m[3] * n[2]));
    }
    return true;
  }
  // imagePoints are normalised (z = 1 plane); RANSAC over 6-point DLT hypotheses, then the motion-only bundle adjustment
  // of the Optimizer path (gh_ba_pnp, Huber at the RANSAC threshold) on the inliers
  bool findPnP(GSLAM::SE3* world2camera, const std::vector<GSLAM::Point3d>& obj, const std::vector<GSLAM::Point2d>& img,
               int method, double threshold, double confidence, std::vector<uchar>* mask) const override {
    if (obj.size() != img.size() || !context()) return false;
    double m[12];
    std::vector<uchar> local;
    if (!estimate(GH_MODEL_PNP, (const double*)obj.data(), (const double*)img.data(), (int)obj.size(), threshold, confidence, m, &local,
                  method))
      return false;
    std::vector<double> X, uv;
    for (size_t i = 0; i < local.size(); ++i)
      if (local[i]) {
        X.push_back(obj[i].x); X.push_back(obj[i].y); X.push_back(obj[i].z);
        uv.push_back(img[i].x); uv.push_back(img[i].y);
      }
    // [R | t] world -> camera  =>  T_wc = (R^T, -R^T t), the parametrisation gh_ba_pnp refines
    GSLAM::SO3 rcw;
    rcw.fromMatrix(m);
    const GSLAM::SE3 Tcw(rcw, GSLAM::Point3d(m[9], m[10], m[11])), Twc = Tcw.inverse();
    const GSLAM::SO3 r = Twc.get_rotation();
    const GSLAM::Point3d t = Twc.get_translation();
    double pose[7] = {r.x, r.y, r.z, r.w, t.x, t.y, t.z};
    gh_ba_options o;
    gh_ba_default_options(&o);
    o.huber_delta = threshold;
    o.max_iterations = 30;
    {
      std::lock_guard<std::mutex> lock(mu_);
      // GH_ERR_NUMERIC = LM stopped on trust-region underflow with a VALID pose (the DLT start was already at the optimum
      // and no step could lower the cost): gh_ba_pnp passes it through, and so does this caller
      const gh_status st = gh_ba_pnp(ctx_, X.data(), uv.data(), (int)(uv.size() / 2), pose, GH_KF_SE3, &o, NULL, NULL);
      if (st != GH_OK && st != GH_ERR_NUMERIC) {
        LOG(ERROR) << "EstimatorHIP: " << gh_last_error(ctx_);
        return false;
      }
    }
    if (world2camera)
      *world2camera = GSLAM::SE3(GSLAM::SO3(pose[0], pose[1], pose[2], pose[3]), GSLAM::Point3d(pose[4], pose[5], pose[6])).inverse();
    if (mask) mask->swap(local);
    return true;
  }
  bool trianglate(GSLAM::Point3d* refPt, const GSLAM::SE3& ref2cur, const GSLAM::Point3d& refDirection,
                  const GSLAM::Point3d& curDirection) const override {
    if (!context()) return false;
    const GSLAM::SO3 r = ref2cur.get_rotation();
    const GSLAM::Point3d t = ref2cur.get_translation();
    const double pose[7] = {r.x, r.y, r.z, r.w, t.x, t.y, t.z};
    const double d1[3] = {refDirection.x, refDirection.y, refDirection.z}, d2[3] = {curDirection.x, curDirection.y, curDirection.z};
    double X[3];
    uint8_t ok = 0;
    std::lock_guard<std::mutex> lock(mu_);
    if (gh_triangulate(ctx_, pose, 0, d1, d2, 1, X, &ok) != GH_OK || !ok) return false;
    if (refPt) *refPt = GSLAM::Point3d(X[0], X[1], X[2]);
    return true;
  }

 private:
  static_assert(sizeof(GSLAM::Point2d) == 16 && sizeof(GSLAM::Point3d) == 24, "point arrays are passed as packed doubles");
  bool run(int model, const std::vector<GSLAM::Point2d>& a, const std::vector<GSLAM::Point2d>& b, int method,
           double threshold, double confidence, double* m, std::vector<uchar>* mask) const {
    if (a.size() != b.size() || !context()) return false;
    return estimate(model, (const double*)a.data(), (const double*)b.data(), (int)a.size(), threshold, confidence, m, mask, method);
  }
  // `confidence` is honoured as a sequential RANSAC would (gh_ransac_estimate_conf): the hypotheses are scored in parallel,
  // the winner is the best of the prefix the adaptive stopping rule would have examined
  static int sampling_of(int model, int method) {
    const std::string forced = svar.GetString("EstimatorHIP.Sampling", "");
    if (forced == "LMEDS") return GH_SAMPLE_LMEDS;
    if (forced == "NOSAMPLE") return GH_SAMPLE_NONE;
    if (forced == "RANSAC") return GH_SAMPLE_RANSAC;
    if ((method & GSLAM::NOSAMPLE) != 0) return GH_SAMPLE_NONE;
    // (GSLAM::LMEDS == F8_Point numerically: for the two-view models the value names the algorithm, not the sampling)
    if (method == GSLAM::LMEDS && model != GH_MODEL_FUNDAMENTAL && model != GH_MODEL_ESSENTIAL) return GH_SAMPLE_LMEDS;
    return GH_SAMPLE_RANSAC;
  }
  bool estimate(int model, const double* a, const double* b, int n, double threshold, double confidence, double* m,
                std::vector<uchar>* mask, int method = 0) const {
    std::lock_guard<std::mutex> lock(mu_);
    std::vector<uchar> local((size_t)(n > 0 ? n : 1));
    int inliers = 0;
    const uint64_t seed = (uint64_t)svar.GetInt("EstimatorHIP.Seed", 1);
    if (gh_ransac_estimate_ex(ctx_, model, a, b, n, threshold, confidence, seed, sampling_of(model, method), m, local.data(), &inliers,
                              NULL) != GH_OK) {
      LOG(ERROR) << "EstimatorHIP: " << gh_last_error(ctx_);
      return false;
    }
    if (mask) mask->assign(local.begin(), local.begin() + n);
    return inliers > 0;
  }
  bool context() const {
    std::lock_guard<std::mutex> lock(mu_);
    if (!ctx_ && gh_ctx_create(svar.GetInt("EstimatorHIP.Device", 0), &ctx_) != GH_OK) {
      ctx_ = nullptr;
      LOG(ERROR) << "EstimatorHIP: no usable HIP device (there is no CPU fallback)";
    }
    return ctx_ != nullptr;
  }
  mutable gh_ctx* ctx_;
  mutable std::mutex mu_;
};

}  // namespace

using GSLAM::funcCreateEstimatorInstance;  // the reference's macro names it unqualified
USE_ESTIMATOR_PLUGIN(EstimatorHIP);
