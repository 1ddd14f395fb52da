// libgslamDB_tumrgbd.so — TUM-RGBD sequence reader (BASELINE configs[0]: "TUM-RGBD fr1_360 mono ... 640x480") on the
// stb path.  The reference's own reader (GSLAM/plugins/datasets/DatasetTUMRGBD.cpp:21-130) is compiled out without
// OpenCV (`#ifdef HAS_OPENCV` around the whole file, cv::imread inside); this plugin reads the same on-disk layout with
// the reference's OpenCV-free image loader (GSLAM/plugins/datasets/IO.h:76-114 -> stb_image) and hands out the
// reference's own frame class (FrameRGBD, GSLAM/plugins/datasets/VideoFrame.h:32-49), so `gslam play -dataset
// <dir>/rgbd.tumrgbd` works on a box without OpenCV.  A caller either side of the hot path: host I/O
// only, nothing here computes on the GPU.
//
// Layout (as the reference's reader expects it, DatasetTUMRGBD.cpp:76-92): <dir>/associate.txt, one line per frame
//     t_pose  tx ty tz qx qy qz qw  t_depth  depth/<file>.png  t_rgb  rgb/<file>.png
// (the output of TUM's associate.py over groundtruth.txt, depth.txt and rgb.txt).  Dataset file: `key value` per line,
// all optional: DatasetFolder, VideoFile, VideoSkip, UseRosCamera, Camera "w h fx fy cx cy [k1 k2 p1 p2 k3]"; camera
// defaults as the reference's detectCamera() (ROS default 640x480 525 525 319.5 239.5, or the freiburg1/2/3 calibrations
// with UseRosCamera 0).  (The reference's readers parse this file with Svar::parseFile, which in this snapshot only
// knows .json / .xml / .yaml / .cfg -- Svar.h:2708-2741 -- and leaves the Svar undefined for any other extension, after
// which the first GetString throws: DatasetKITTI.cpp:33-36.  A plain key-value file has no such dependency.)
#include <GSLAM/core/GSLAM.h>

#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <vector>

#include "GSLAM/plugins/datasets/IO.h"
#include "GSLAM/plugins/datasets/VideoFrame.h"

using namespace GSLAM;

namespace {

class DatasetTUMRGBDStb : public Dataset {
 public:
  DatasetTUMRGBDStb() : skip_(0), next_id_(1) {}
  std::string type() const override { return "DatasetTUMRGBD"; }
  bool isOpened() override { return camera_.isValid() && ifs_.is_open(); }

  bool open(const std::string& dataset) override {
    std::map<std::string, std::string> kv;
    {
      std::ifstream f(dataset.c_str());
      if (!f.is_open()) return false;
      std::string line;
      while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string k, v;
        if (!(ss >> k) || k[0] == '#') continue;
        std::getline(ss, v);
        const size_t b = v.find_first_not_of(" \t");
        kv[k] = b == std::string::npos ? std::string() : v.substr(b);
      }
    }
    std::string folder = dataset;
    const size_t slash = folder.find_last_of("/\\");
    folder = slash == std::string::npos ? std::string(".") : folder.substr(0, slash);
    top_ = kv.count("DatasetFolder") ? kv["DatasetFolder"] : folder;
    skip_ = kv.count("VideoSkip") ? atoi(kv["VideoSkip"].c_str()) : 0;
    if (kv.count("Camera")) {
      std::istringstream ss(kv["Camera"]);
      std::vector<double> p;
      double v;
      while (ss >> v) p.push_back(v);
      camera_ = Camera(p);
    }
    if (!camera_.isValid()) camera_ = default_camera(top_, kv.count("UseRosCamera") ? atoi(kv["UseRosCamera"].c_str()) != 0 : true);
    if (!camera_.isValid()) {
      LOG(ERROR) << "DatasetTUMRGBD(stb): camera not valid: " << camera_.info();
      return false;
    }
    const std::string assoc = kv.count("VideoFile") ? kv["VideoFile"] : top_ + "/associate.txt";
    ifs_.open(assoc.c_str());
    if (!ifs_.is_open()) {
      LOG(ERROR) << "DatasetTUMRGBD(stb): cannot open " << assoc;
      return false;
    }
    _name = dataset;
    return true;
  }

  FramePtr grabFrame() override {
    std::string line;
    double t_pose = 0, t_depth = 0, t_rgb = 0, p[7];
    std::string depth_file, rgb_file;
    for (int i = 0; i < skip_ + 1; ++i) {
      do {
        if (!std::getline(ifs_, line)) return FramePtr();
      } while (line.empty() || line[0] == '#');
      std::istringstream ss(line);
      ss >> t_pose >> p[0] >> p[1] >> p[2] >> p[3] >> p[4] >> p[5] >> p[6] >> t_depth >> depth_file >> t_rgb >> rgb_file;
      if (!ss) return FramePtr();
    }
    GImage img = imread(top_ + "/" + rgb_file);  // stb -> BGR(A) / gray (IO.h:100-113)
    if (img.empty()) return FramePtr();
    GImage depth = read_depth(top_ + "/" + depth_file);
    FramePtr frame(new FrameRGBD(img, depth, camera_, next_id_++, t_rgb));
    frame->setPose(SE3(SO3(p[3], p[4], p[5], p[6]), Point3d(p[0], p[1], p[2])));  // TUM ground truth: tx ty tz qx qy qz qw
    return frame;
  }

 private:
  // 16-bit depth PNGs (factor 5000 in the TUM sets) keep their 16 bits: stbi_load would squeeze them to 8
  static GImage read_depth(const std::string& path) {
    int x = 0, y = 0, ch = 0;
    // (this stb_image has no stbi_is_16_bit: the bit depth is byte 24 of a PNG, right after the IHDR width / height)
    unsigned char head[26] = {0};
    {
      std::ifstream f(path.c_str(), std::ios::binary);
      f.read((char*)head, sizeof(head));
    }
    const bool png16 = head[1] == 'P' && head[2] == 'N' && head[3] == 'G' && head[24] == 16;
    if (png16) {
      stbi_us* d = stbi_load_16(path.c_str(), &x, &y, &ch, 1);
      if (!d) return GImage();
      GImage out(y, x, GImageType<uint16_t, 1>::Type, (uchar*)d, true);
      stbi_image_free(d);
      return out;
    }
    return imread(path);
  }
  static Camera default_camera(const std::string& top, bool ros_default) {
    if (ros_default) return Camera({640, 480, 525.0, 525.0, 319.5, 239.5});
    const size_t idx = top.find("freiburg");
    const char c = idx == std::string::npos || idx + 8 >= top.size() ? '0' : top[idx + 8];
    if (c == '1') return Camera({640, 480, 517.3, 516.5, 318.6, 255.3, 0.2624, -0.9531, -0.0054, 0.0026, 1.1633});
    if (c == '2') return Camera({640, 480, 520.9, 521.0, 325.1, 249.7, 0.2312, -0.7849, -0.0033, -0.0001, 0.9172});
    if (c == '3') return Camera({640, 480, 535.4, 539.2, 320.1, 247.6});
    return Camera({640, 480, 525.0, 525.0, 319.5, 239.5});
  }

  std::string top_;
  Camera camera_;
  std::ifstream ifs_;
  int skip_;
  FrameID next_id_;
};

}  // namespace

GSLAM_REGISTER_DATASET(DatasetTUMRGBDStb, tumrgbd)
