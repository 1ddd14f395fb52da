// libgslam_estimator.so — GSLAM::Estimator plugin (GSLAM/core/Estimator.h:92-191) on the MI355X.
// Exports `createEstimatorInstance` through the reference's own USE_ESTIMATOR_PLUGIN macro (:42-53), so
// GSLAM::Estimator::create() (default plugin name "libgslam_estimator", svar EstimatorPlugin) loads it unchanged.
// RANSAC with inlier masks for findHomography / findAffine2D / findFundamental / findAffine3D -> gh_ransac_estimate.
// The other pure virtuals (findEssentialMatrix, findSIM3, findPlane, findPnP, trianglate) and the NOSAMPLE mode return
// false ("unsupported"), as callers of the interface must already expect from the bool result.
// Note on `method`: in the reference's EstimatorMethod enum the model ids follow MODEL_METHOD = 0xFF, i.e. they are
// 0x100..0x111 and collide with the sampling bits (LMEDS = 1 << 8); the interface's own defaults are `X & RANSAC` = 0.
// The only sampling flag that can be told apart is NOSAMPLE (2 << 8), so every other value means RANSAC here.
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
    if (!run(GH_MODEL_HOMOGRAPHY, src, dst, method, threshold, m, mask)) return false;
    if (H) for (int i = 0; i < 9; ++i) H->data()[i] = m[i];
    return true;
  }
  bool findAffine2D(GSLAM::Affine2D* A, const std::vector<GSLAM::Point2d>& src, const std::vector<GSLAM::Point2d>& dst,
                    int method, double threshold, double confidence, std::vector<uchar>* mask) const override {
    double m[12];
    if (!run(GH_MODEL_AFFINE2D, src, dst, method, threshold, m, mask)) return false;
    if (A) for (int i = 0; i < 6; ++i) A->data()[i] = m[i];
    return true;
  }
  bool findFundamental(GSLAM::Fundamental* F, const std::vector<GSLAM::Point2d>& p1,
                       const std::vector<GSLAM::Point2d>& p2, int method, double threshold, double confidence,
                       std::vector<uchar>* mask) const override {
    double m[12];
    if (!run(GH_MODEL_FUNDAMENTAL, p1, p2, method, threshold, m, mask)) return false;
    if (F) for (int i = 0; i < 9; ++i) F->data()[i] = m[i];
    return true;
  }
  bool findAffine3D(GSLAM::Affine3D* A, const std::vector<GSLAM::Point3d>& src, const std::vector<GSLAM::Point3d>& dst,
                    int method, double threshold, double confidence, std::vector<uchar>* mask) const override {
    if (src.size() != dst.size() || (method & GSLAM::NOSAMPLE) != 0 || !context()) return false;
    double m[12];
    if (!estimate(GH_MODEL_AFFINE3D, (const double*)src.data(), (const double*)dst.data(), (int)src.size(), threshold, m,
                  mask))
      return false;
    if (A) for (int i = 0; i < 12; ++i) A->data()[i] = m[i];
    return true;
  }
  bool findEssentialMatrix(GSLAM::Essential*, const std::vector<GSLAM::Point2d>&, const std::vector<GSLAM::Point2d>&,
                           int, double, double, std::vector<uchar>*) const override { return false; }
  bool findSIM3(GSLAM::SIM3*, const std::vector<GSLAM::Point3d>&, const std::vector<GSLAM::Point3d>&, int, double,
                double, std::vector<uchar>*) const override { return false; }
  bool findPlane(GSLAM::SE3*, const std::vector<GSLAM::Point3d>&, int, double, double,
                 std::vector<uchar>*) const override { return false; }
  bool findPnP(GSLAM::SE3*, const std::vector<GSLAM::Point3d>&, const std::vector<GSLAM::Point2d>&, int, double, double,
               std::vector<uchar>*) const override { return false; }
  bool trianglate(GSLAM::Point3d*, const GSLAM::SE3&, const GSLAM::Point3d&, const GSLAM::Point3d&) const override {
    return false;
  }

 private:
  static_assert(sizeof(GSLAM::Point2d) == 16 && sizeof(GSLAM::Point3d) == 24, "point arrays are passed as packed doubles");
  bool run(int model, const std::vector<GSLAM::Point2d>& a, const std::vector<GSLAM::Point2d>& b, int method,
           double threshold, double* m, std::vector<uchar>* mask) const {
    if (a.size() != b.size() || (method & GSLAM::NOSAMPLE) != 0 || !context()) return false;
    return estimate(model, (const double*)a.data(), (const double*)b.data(), (int)a.size(), threshold, m, mask);
  }
  bool estimate(int model, const double* a, const double* b, int n, double threshold, double* m,
                std::vector<uchar>* mask) const {
    std::lock_guard<std::mutex> lock(mu_);
    std::vector<uchar> local((size_t)(n > 0 ? n : 1));
    int inliers = 0;
    const uint64_t seed = (uint64_t)svar.GetInt("EstimatorHIP.Seed", 1);
    if (gh_ransac_estimate(ctx_, model, a, b, n, threshold, seed, m, local.data(), &inliers) != GH_OK) {
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
