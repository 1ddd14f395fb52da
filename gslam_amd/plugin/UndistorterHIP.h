// UndistorterHIP — drop-in for GSLAM::Undistorter (GSLAM/core/Undistorter.h:90-118) whose per-pixel remap runs on
// the MI355X.  GSLAM::Undistorter is a concrete pimpl class (no virtuals, no factory), so the replacement is a class
// with the SAME public methods: construct from (Camera in, Camera out), undistort / undistortFast (two-argument and
// value-returning forms), cameraIn / cameraOut, prepareReMap, valid.  The remap tables are built by the reference's
// own UndistorterImpl::prepareReMap (host-side camera maths stay in GSLAM); only the image loops move to
// gh_undistort_host (include/gslam_hip.h).  Header only; link with -lgslam_hip.
#ifndef GSLAM_AMD_UNDISTORTER_HIP_H_
#define GSLAM_AMD_UNDISTORTER_HIP_H_

#include <GSLAM/core/GSLAM.h>
#include <GSLAM/core/Undistorter.h>

#include <iostream>
#include <memory>

#include "gslam_hip.h"

namespace GSLAM {

class UndistorterHIP {
 public:
  UndistorterHIP(Camera in = Camera(), Camera out = Camera()) : ctx_(nullptr), plan_(nullptr) {
    std::streambuf* old = std::cout.rdbuf(nullptr);  // prepareReMap prints the camera info
    impl_ = std::shared_ptr<UndistorterImpl>(new UndistorterImpl(in, out));
    std::cout.rdbuf(old);
    upload();
  }
  ~UndistorterHIP() {
    if (plan_) gh_undist_plan_destroy(plan_);
    if (ctx_) gh_ctx_destroy(ctx_);
  }
  UndistorterHIP(const UndistorterHIP&) = delete;
  UndistorterHIP& operator=(const UndistorterHIP&) = delete;

  bool undistort(const GImage& image, GImage& result) { return run(image, result, 0); }
  bool undistortFast(const GImage& image, GImage& result) { return run(image, result, 1); }
  GImage undistort(const GImage& image) {
    GImage r;
    return undistort(image, r) ? r : GImage();
  }
  GImage undistortFast(const GImage& image) {
    GImage r;
    return undistortFast(image, r) ? r : GImage();
  }
  Camera cameraIn() { return impl_->camera_in; }
  Camera cameraOut() { return impl_->camera_out; }
  bool prepareReMap() { return impl_->prepareReMap() && upload(); }
  bool valid() { return impl_->valid && plan_ != nullptr; }

 private:
  bool upload() {
    if (!impl_->valid) return false;
    if (!ctx_ && gh_ctx_create(svar.GetInt("UndistorterHIP.Device", 0), &ctx_) != GH_OK) {
      ctx_ = nullptr;
      LOG(ERROR) << "UndistorterHIP: no usable HIP device (there is no CPU fallback)";
      return false;
    }
    if (plan_) gh_undist_plan_destroy(plan_);
    plan_ = nullptr;
    return gh_undist_plan_create(ctx_, impl_->camera_in.width(), impl_->camera_in.height(), impl_->camera_out.width(),
                                 impl_->camera_out.height(), impl_->remapX, impl_->remapFast, impl_->remapIdx,
                                 impl_->remapCoef, &plan_) == GH_OK;
  }
  bool run(const GImage& image, GImage& result, int fast) {
    if (!valid()) {  // same contract as the reference: pass the image through and report failure
      result = image;
      return false;
    }
    if (image.rows != impl_->camera_in.height() || image.cols != impl_->camera_in.width() || image.elemSize1() != 1) {
      result = image;
      return false;
    }
    const int c = image.channels();
    if (!(c == 1 || c == 3 || (c == 4 && fast))) {
      result = image;
      return false;
    }
    result = GImage(impl_->camera_out.height(), impl_->camera_out.width(), image.type());
    return gh_undistort_host(plan_, image.data, c, result.data, fast) == GH_OK;
  }
  std::shared_ptr<UndistorterImpl> impl_;
  gh_ctx* ctx_;
  gh_undist_plan* plan_;
};

}  // namespace GSLAM
#endif  // GSLAM_AMD_UNDISTORTER_HIP_H_
