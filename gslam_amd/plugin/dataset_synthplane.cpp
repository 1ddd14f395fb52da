// libgslamDB_synthplane.so — a GSLAM *dataset* plugin (GSLAM_REGISTER_DATASET, GSLAM/core/GSLAM.h:35-42) that feeds the
// reference's own `play` application (GSLAM/plugins/play/main.cpp) with a synthetic monocular sequence, the stand-in for
// BASELINE configs[0] (TUM-RGBD fr1_360, 640x480, ROS-default TUM pinhole): no dataset, image file or OpenCV exists
// offline, so the sequence is rendered here.
//
//   `gslam play -dataset seq.synthplane -autostart 1 orbhip metric_time -slam orbhip`
//       Dataset::open("seq.synthplane") -> Registry::load("gslamDB_synthplane") (GSLAM/core/Dataset.h:124-162)
//
// Scene: the plane z = 0 carrying a T x T procedural texture (the corner-rich integer texture of the bench frames,
// generated in HBM by gh_synth_frames_dev and downloaded once); camera: TUM default pinhole
// {640, 480, 525, 525, 319.5, 239.5} (GSLAM/plugins/datasets/DatasetTUMRGBD.cpp:106) on a smooth orbit looking at the
// plane.  A frame is rendered by intersecting every pixel ray with the plane and sampling the texture bilinearly; it is
// delivered as a BGR FrameMono-style frame (datasets deliver BGR/BGRA, GSLAM/plugins/datasets/IO.h:100-113) whose pose
// is the ground truth T_wc.  Geometrically consistent, so extract -> match -> optimizePnP -> optimize tracks it.
//
// File format (text, `key value` per line): width height fx fy cx cy frames texture scale radius height seed fps
//   [dump <path>]  raw record of what was delivered, for the tests: header {int32 n, w, h}, then per frame
//                  {int32 id, double t, double pose[7] (qx qy qz qw tx ty tz), w*h gray bytes}.
#include <GSLAM/core/GSLAM.h>

#include <cmath>
#include <cstdint>
#include <fstream>
#include <map>
#include <sstream>
#include <vector>

#include "gslam_hip.h"

using namespace GSLAM;

namespace {

// What the reference's FrameMono (GSLAM/plugins/datasets/VideoFrame.h:14-30,162-181) is, plus the keypoint storage the
// front end deposits through MapFrame::setKeyPoints (Map.h:309-321; the gmap plugin's frame does the same,
// plugins/gmap/MapFrame.cpp:211-247).
class FramePlane : public MapFrame {
 public:
  FramePlane(FrameID id, double t, const GImage& bgr, const GImage& gray, const Camera& cam)
      : MapFrame(id, t), bgr_(bgr), gray_(gray), cam_(cam) {}
  std::string type() const override { return "FrameMono"; }
  int cameraNum() const override { return 1; }
  int imageChannels(int) const override { return IMAGE_BGRA; }
  GImage getImage(int, int channels) override { return (channels == IMAGE_GRAY) ? gray_ : bgr_; }
  Camera getCamera(int) override { return cam_; }
  int keyPointNum() const override { return (int)kps_.size(); }
  bool setKeyPoints(const std::vector<KeyPoint>& k, const GImage& d) override {
    WriteMutex lock(_mutexPose);
    kps_ = k;
    desc_ = d.clone();
    return true;
  }
  bool getKeyPoints(std::vector<KeyPoint>& k) const override {
    ReadMutex lock(_mutexPose);
    k = kps_;
    return true;
  }
  bool getKeyPoint(int idx, KeyPoint& pt) const override {
    ReadMutex lock(_mutexPose);
    if (idx < 0 || idx >= (int)kps_.size()) return false;
    pt = kps_[idx];
    return true;
  }
  GImage getDescriptor(int idx) const override { return idx < 0 ? desc_ : desc_.row(idx); }

 private:
  GImage bgr_, gray_, desc_;
  Camera cam_;
  std::vector<KeyPoint> kps_;
};

class DatasetSynthPlane : public Dataset {
 public:
  DatasetSynthPlane() : next_(0), opened_(false) {}
  ~DatasetSynthPlane() override {
    if (dump_.is_open()) dump_.close();
  }
  std::string type() const override { return "DatasetSynthPlane"; }
  bool isOpened() override { return opened_; }

  bool open(const std::string& path) override {
    std::ifstream f(path.c_str());
    if (!f.is_open()) return false;
    std::map<std::string, std::string> kv;
    std::string k, v;
    while (f >> k >> v) kv[k] = v;
    auto num = [&](const char* key, double def) { return kv.count(key) ? atof(kv[key].c_str()) : def; };
    w_ = (int)num("width", 640);
    h_ = (int)num("height", 480);
    fx_ = num("fx", 525);
    fy_ = num("fy", 525);
    cx_ = num("cx", 319.5);
    cy_ = num("cy", 239.5);
    n_ = (int)num("frames", 300);
    T_ = (int)num("texture", 2048);
    S_ = num("scale", 4.0);        // the texture covers [-S, S]^2 of the plane
    radius_ = num("radius", 0.8);  // orbit radius of the camera centre
    height_ = num("height_m", 2.2);
    fps_ = num("fps", 30);
    const uint32_t seed = (uint32_t)num("seed", 0x5EED0000);
    if (w_ <= 0 || h_ <= 0 || n_ <= 0 || T_ < 64) return false;
    camera_ = Camera(std::vector<double>({(double)w_, (double)h_, fx_, fy_, cx_, cy_}));
    if (!camera_.isValid()) return false;
    // texture: one T x T frame of the procedural generator, made on the GPU and downloaded once
    gh_ctx* ctx = nullptr;
    if (gh_ctx_create(svar.GetInt("DatasetSynthPlane.Device", 0), &ctx) != GH_OK) {
      LOG(ERROR) << "DatasetSynthPlane: no usable HIP device";
      return false;
    }
    tex_.resize((size_t)T_ * T_);
    void* d = nullptr;
    bool ok = gh_dev_alloc(ctx, tex_.size(), &d) == GH_OK &&
              gh_synth_frames_dev(ctx, (uint8_t*)d, T_, T_, T_, (size_t)T_ * T_, 0, 1, seed) == GH_OK &&
              gh_dev_download(ctx, tex_.data(), d, tex_.size()) == GH_OK;
    if (d) gh_dev_free(ctx, d);
    gh_ctx_destroy(ctx);
    if (!ok) return false;
    if (kv.count("dump")) {
      dump_.open(kv["dump"].c_str(), std::ios::binary);
      int32_t hdr[3] = {n_, w_, h_};
      dump_.write((const char*)hdr, sizeof(hdr));
    }
    _name = path;
    next_ = 0;
    opened_ = true;
    return true;
  }

  // T_wc of frame i: the centre moves on a circle of radius `radius` at height `height` and keeps looking at a point
  // that itself wanders slowly over the plane (so rotation AND translation change every frame)
  void pose_of(int i, double* q, double* t) const {
    const double a = 2.0 * M_PI * i / (double)(n_ > 120 ? 120 : n_);
    const double C[3] = {radius_ * cos(a), radius_ * sin(a), height_ + 0.15 * sin(0.5 * a)};
    const double L[3] = {0.3 * cos(0.37 * a + 1.0), 0.3 * sin(0.23 * a), 0.0};
    double z[3] = {L[0] - C[0], L[1] - C[1], L[2] - C[2]};
    const double zn = sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
    for (int e = 0; e < 3; ++e) z[e] /= zn;
    const double up[3] = {0, 1, 0};
    double x[3] = {up[1] * z[2] - up[2] * z[1], up[2] * z[0] - up[0] * z[2], up[0] * z[1] - up[1] * z[0]};
    const double xn = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    for (int e = 0; e < 3; ++e) x[e] /= xn;
    const double y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
    // R = [x y z] (columns = camera axes in the world) -> quaternion
    const double R[9] = {x[0], y[0], z[0], x[1], y[1], z[1], x[2], y[2], z[2]};
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
      const double s = sqrt(tr + 1.0) * 2;
      q[3] = 0.25 * s; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
      const double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
      q[3] = (R[7] - R[5]) / s; q[0] = 0.25 * s; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
      const double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
      q[3] = (R[2] - R[6]) / s; q[0] = (R[1] + R[3]) / s; q[1] = 0.25 * s; q[2] = (R[5] + R[7]) / s;
    } else {
      const double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
      q[3] = (R[3] - R[1]) / s; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = 0.25 * s;
    }
    t[0] = C[0]; t[1] = C[1]; t[2] = C[2];
  }

  FramePtr grabFrame() override {
    if (!opened_ || next_ >= n_) return FramePtr();
    const int i = next_++;
    double q[4], t[3];
    pose_of(i, q, t);
    const SE3 Twc(SO3(q[0], q[1], q[2], q[3]), Point3d(t[0], t[1], t[2]));
    std::vector<uint8_t> gray((size_t)w_ * h_), bgr((size_t)w_ * h_ * 3);
    const double scale = T_ / (2.0 * S_);
    for (int v = 0; v < h_; ++v)
      for (int u = 0; u < w_; ++u) {
        const Point3d dir = Twc.get_rotation() * Point3d((u - cx_) / fx_, (v - cy_) / fy_, 1.0);
        uint8_t g = 0;
        if (dir.z < -1e-9) {
          const double lam = -t[2] / dir.z;
          const double tx = (t[0] + lam * dir.x + S_) * scale - 0.5, ty = (t[1] + lam * dir.y + S_) * scale - 0.5;
          const int x0 = (int)floor(tx), y0 = (int)floor(ty);
          if (x0 >= 0 && y0 >= 0 && x0 + 1 < T_ && y0 + 1 < T_) {
            const double ax = tx - x0, ay = ty - y0;
            const uint8_t* p = &tex_[(size_t)y0 * T_ + x0];
            const double val = (p[0] * (1 - ax) + p[1] * ax) * (1 - ay) + (p[T_] * (1 - ax) + p[T_ + 1] * ax) * ay;
            g = (uint8_t)(val + 0.5);
          }
        }
        gray[(size_t)v * w_ + u] = g;
        uint8_t* c = &bgr[((size_t)v * w_ + u) * 3];
        c[0] = c[1] = c[2] = g;
      }
    const double stamp = i / fps_;
    if (dump_.is_open()) {
      const int32_t id = i + 1;
      const double pose[7] = {q[0], q[1], q[2], q[3], t[0], t[1], t[2]};
      dump_.write((const char*)&id, 4);
      dump_.write((const char*)&stamp, 8);
      dump_.write((const char*)pose, sizeof(pose));
      dump_.write((const char*)gray.data(), gray.size());
      dump_.flush();
    }
    GImage gbgr(h_, w_, GImageType<uchar, 3>::Type, bgr.data(), true);
    GImage ggray(h_, w_, GImageType<uchar, 1>::Type, gray.data(), true);
    FramePtr fr(new FramePlane(i + 1, stamp, gbgr, ggray, camera_));
    fr->setPose(Twc);
    return fr;
  }

 private:
  int w_, h_, n_, T_, next_;
  double fx_, fy_, cx_, cy_, S_, radius_, height_, fps_;
  bool opened_;
  Camera camera_;
  std::vector<uint8_t> tex_;
  std::ofstream dump_;
};

}  // namespace

GSLAM_REGISTER_DATASET(DatasetSynthPlane, synthplane)
