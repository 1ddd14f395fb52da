// libgslamDB_kitti.so — KITTI odometry sequence reader (BASELINE configs[2]: "KITTI 00 stereo") on the stb path.
//
// Why not the reference's own: GSLAM/plugins/datasets/DatasetKITTI.cpp does have an OpenCV-free image path (IO.h:76-114)
// and compiles here, but its open() starts with `Svar var; var.parseFile(dataset); var.GetString(...)` (:31-36) and
// Svar::loadFile only knows .json / .xml / .yaml / .cfg (Svar.h:2708-2741): for "<dir>/stereo.kitti" the Svar stays
// undefined and the first GetString throws "Unable cast void to str" -- the reader cannot open anything in this snapshot.
// This plugin reads the same layout and hands out the reference's own frame classes (FrameMono / FrameStereo,
// GSLAM/plugins/datasets/VideoFrame.h), with the same calibration convention (DatasetKITTI.cpp:116-135):
//     <dir>/image_<i>/%06d.png (i = 0..3, whichever exist), <dir>/times.txt, <dir>/calib.txt (P0..P3, 12 numbers each),
//     <dir>/pose.txt (optional ground truth, 12 numbers per line = 3 x 4 T_wc)
// Dataset file `<dir>/<VideoType>.kitti` (`key value` lines, all optional): SequenceFolder, VideoType (mono | stereo,
// default = the file's base name), CameraIdx, GroundFile.  Host I/O only.
#include <GSLAM/core/GSLAM.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <vector>

#include "GSLAM/plugins/datasets/IO.h"
#include "GSLAM/plugins/datasets/VideoFrame.h"

using namespace GSLAM;

namespace {

class DatasetKITTIStb : public Dataset {
 public:
  DatasetKITTIStb() : stereo_(false), cam_idx_(0), cur_(0), mask_(0) {}
  std::string type() const override { return "DatasetKITTI"; }
  bool isOpened() override { return !stamps_.empty(); }

  bool open(const std::string& dataset) override {
    std::map<std::string, std::string> kv;
    {
      std::ifstream f(dataset.c_str());
      if (!f.is_open()) return false;
      std::string k, v;
      while (f >> k >> v) kv[k] = v;
    }
    const size_t slash = dataset.find_last_of("/\\");
    const std::string dir = slash == std::string::npos ? std::string(".") : dataset.substr(0, slash);
    std::string base = slash == std::string::npos ? dataset : dataset.substr(slash + 1);
    base = base.substr(0, base.find_last_of('.'));
    folder_ = kv.count("SequenceFolder") ? kv["SequenceFolder"] : dir;
    const std::string type = kv.count("VideoType") ? kv["VideoType"] : base;
    cam_idx_ = kv.count("CameraIdx") ? atoi(kv["CameraIdx"].c_str()) : 0;
    // calibration: P_i = K [I | t_i]: fx = p0, fy = p5, cx = p2, cy = p6, baseline from p3 (DatasetKITTI.cpp:116-135)
    std::ifstream calib((folder_ + "/calib.txt").c_str());
    if (!calib.is_open()) {
      LOG(ERROR) << "DatasetKITTI(stb): cannot open " << folder_ << "/calib.txt";
      return false;
    }
    std::string line;
    for (int i = 0; i < 4 && std::getline(calib, line); ++i) {
      std::istringstream ss(line.substr(line.find(':') == std::string::npos ? 0 : line.find(':') + 1));
      double p[12];
      for (int k = 0; k < 12; ++k)
        if (!(ss >> p[k])) return false;
      cam_[i] = Camera({1241, 376, p[0], p[5], p[2], p[6]});
      pos_[i] = SE3(SO3(), Point3d(-p[3] / p[0], -p[7] / p[5], -p[11]));
    }
    mask_ = 0;
    for (int i = 0; i < 4; ++i) {
      char name[64];
      snprintf(name, sizeof(name), "/image_%d/000000.png", i);
      if (!imread(folder_ + name).empty()) mask_ |= 1 << i;
    }
    if (!mask_) {
      LOG(ERROR) << "DatasetKITTI(stb): no image_<i>/000000.png under " << folder_;
      return false;
    }
    stereo_ = !(type == "mono" || type == "Mono" || type == "Monocular");
    if (!stereo_ && !(mask_ & (1 << cam_idx_)))
      for (cam_idx_ = 0; !(mask_ & (1 << cam_idx_)); ++cam_idx_) {}
    if (stereo_ && __builtin_popcount(mask_) < 2) {
      LOG(ERROR) << "DatasetKITTI(stb): stereo needs two image folders";
      return false;
    }
    std::ifstream times((folder_ + "/times.txt").c_str());
    if (!times.is_open()) {
      LOG(ERROR) << "DatasetKITTI(stb): cannot open " << folder_ << "/times.txt";
      return false;
    }
    while (std::getline(times, line))
      if (!line.empty()) stamps_.push_back(atof(line.c_str()));
    std::ifstream gt((kv.count("GroundFile") ? kv["GroundFile"] : folder_ + "/pose.txt").c_str());
    while (gt.is_open() && std::getline(gt, line)) {
      std::istringstream ss(line);
      double m[12];
      bool ok = true;
      for (int k = 0; k < 12; ++k) ok = ok && (ss >> m[k]);
      if (!ok) break;
      SE3 T;
      T.fromMatrix(m);
      poses_.push_back(T);
    }
    if (poses_.size() != stamps_.size()) poses_.clear();
    cur_ = 0;
    _name = dataset;
    return !stamps_.empty();
  }

  FramePtr grabFrame() override {
    if (cur_ >= (int)stamps_.size()) return FramePtr();
    char file[32];
    snprintf(file, sizeof(file), "/%06d.png", cur_);
    FramePtr fr;
    if (!stereo_) {
      GImage img = imread(folder_ + "/image_" + std::to_string(cam_idx_) + file);
      if (img.empty()) return FramePtr();
      fr = FramePtr(new FrameMono(cur_, stamps_[cur_], img, cam_[cam_idx_], img.channels() == 1 ? IMAGE_GRAY : IMAGE_BGRA));
    } else {
      std::vector<GImage> imgs;
      std::vector<int> idx;
      for (int i = 0; i < 4 && imgs.size() < 2; ++i) {
        if (!(mask_ & (1 << i))) continue;
        GImage img = imread(folder_ + "/image_" + std::to_string(i) + file);
        if (img.empty()) return FramePtr();
        imgs.push_back(img);
        idx.push_back(i);
      }
      fr = FramePtr(new FrameStereo(imgs[0], imgs[1], cam_[idx[0]], cam_[idx[1]], pos_[idx[0]].inverse() * pos_[idx[1]], cur_,
                                    stamps_[cur_]));
    }
    if (!poses_.empty()) fr->setPose(poses_[cur_]);
    ++cur_;
    return fr;
  }

 private:
  std::string folder_;
  bool stereo_;
  int cam_idx_, cur_, mask_;
  std::vector<double> stamps_;
  std::vector<SE3> poses_;
  Camera cam_[4];
  SE3 pos_[4];
};

}  // namespace

GSLAM_REGISTER_DATASET(DatasetKITTIStb, kitti)
