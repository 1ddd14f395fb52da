// FeatureDetector — plugin interface for the ORB front end + brute-force matcher.
//
// GSLAM 3.0.0 has NO such interface (the ORB extractor lives in the external gslam_orbslam plugin;
// grep of /root/reference finds only LoopDetector, GSLAM/core/Map.h:381-395).  This header adds one in
// the idiom of GSLAM/core/Optimizer.h:42-51,184-253: an abstract class with bool-returning virtuals
// that default to "unsupported", a static create() that resolves `FeatureDetectorPlugin`
// (default "libgslam_featuredetector") through GSLAM::Registry, and an unmangled factory symbol
// `createFeatureDetectorInstance`.  Data types are GSLAM's own: GImage (8UC1 image / N x 32 descriptor
// matrix, GSLAM/core/GImage.h), KeyPoint (GSLAM/core/Map.h:122-195), match list as
// vector<pair<int,int>> (FrameConnection::getMatches, Map.h:252-258), inlier mask vector<uchar>
// (as in GSLAM/core/Estimator.h:100-169).
#ifndef GSLAM_AMD_FEATUREDETECTOR_H_
#define GSLAM_AMD_FEATUREDETECTOR_H_

#include <GSLAM/core/GSLAM.h>

#include <memory>
#include <string>
#include <utility>
#include <vector>

#define GSLAM_REGISTER_FEATUREDETECTOR(DET_CLASS)                                          \
  extern "C" std::shared_ptr<GSLAM::FeatureDetector> createFeatureDetectorInstance() {     \
    return std::shared_ptr<GSLAM::FeatureDetector>(new DET_CLASS());                       \
  }

namespace GSLAM {

class FeatureDetector;
typedef std::shared_ptr<FeatureDetector> (*funcCreateFeatureDetectorInstance)();

struct FeatureDetectorConfig {
  int nFeatures = 1000;   // ORB-SLAM nFeatures
  int nLevels = 8;        // scale factor fixed at 1.2
  int iniThFAST = 20;
  int minThFAST = 7;
  int matchMaxDistance = 100;   // Hamming threshold (ORB-SLAM TH_HIGH)
  int matchRatioNum = 0, matchRatioDen = 1;  // ratio test d1 * den < num * d2; num <= 0 disables
  bool matchCrossCheck = false;
};

class FeatureDetector {
 public:
  explicit FeatureDetector(FeatureDetectorConfig config = FeatureDetectorConfig()) : _config(config) {}
  virtual ~FeatureDetector() {}

  // gray (8UC1) or BGR/BGRA (8UC3/8UC4) image -> keypoints + N x 32 descriptors (8UC1)
  virtual bool detectAndCompute(const GImage& image, std::vector<KeyPoint>& keypoints, GImage& descriptors) {
    return false;
  }
  // brute-force Hamming matches (query idx, train idx) for the queries that pass the configured tests;
  // mask (optional) has one entry per query row
  virtual bool match(const GImage& queryDescriptors, const GImage& trainDescriptors,
                     std::vector<std::pair<int, int> >& matches, std::vector<uchar>* mask = NULL) {
    return false;
  }

  // Many frame pairs in one call (a keyframe against its covisible keyframes, loop candidates, an offline all-pairs pass):
  // descriptors[f] is the N_f x 32 matrix of frame f, pairs[p] = (query frame, train frame); matches[p] as match() returns
  // them for that pair, under the same configured tests.  The base class loops over match(); an accelerated plugin runs all
  // pairs as one batch.
  virtual bool matchBatch(const std::vector<GImage>& descriptors, const std::vector<std::pair<int, int> >& pairs,
                          std::vector<std::vector<std::pair<int, int> > >& matches) {
    matches.assign(pairs.size(), std::vector<std::pair<int, int> >());
    for (size_t p = 0; p < pairs.size(); ++p) {
      if (pairs[p].first < 0 || pairs[p].second < 0 || pairs[p].first >= (int)descriptors.size() || pairs[p].second >= (int)descriptors.size())
        return false;
      if (!match(descriptors[pairs[p].first], descriptors[pairs[p].second], matches[p])) return false;
    }
    return true;
  }

  // A burst of same-sized images in one call (keyframe burst, stereo pair, offline sequence): the base class loops over
  // detectAndCompute, an accelerated plugin pipelines uploads, extraction and downloads.
  virtual bool detectAndComputeBatch(const std::vector<GImage>& images, std::vector<std::vector<KeyPoint> >& keypoints,
                                     std::vector<GImage>& descriptors) {
    keypoints.resize(images.size());
    descriptors.resize(images.size());
    for (size_t i = 0; i < images.size(); ++i)
      if (!detectAndCompute(images[i], keypoints[i], descriptors[i])) return false;
    return true;
  }
  // Asynchronous pair for a caller that receives frames one at a time (GSLAM/plugins/play/main.cpp:99-155 publishes a
  // FramePtr per frame): submit() returns a ticket at once (< 0: unsupported / failed) while the image is still on its
  // way to the device, collect() blocks until that frame's records are on the host.  At most asyncDepth() tickets may be
  // outstanding; tickets are collected in any order.
  virtual long submit(const GImage& image) { return -1; }
  virtual bool collect(long ticket, std::vector<KeyPoint>& keypoints, GImage& descriptors) { return false; }
  virtual int asyncDepth() const { return 0; }

  static std::shared_ptr<FeatureDetector> create(std::string pluginName = "") {
    if (pluginName.empty()) pluginName = svar.GetString("FeatureDetectorPlugin", "libgslam_featuredetector");
    std::shared_ptr<SharedLibrary> plugin = Registry::get(pluginName);
    if (!plugin) return std::shared_ptr<FeatureDetector>();
    funcCreateFeatureDetectorInstance f =
        (funcCreateFeatureDetectorInstance)plugin->getSymbol("createFeatureDetectorInstance");
    if (!f) return std::shared_ptr<FeatureDetector>();
    return f();
  }

  FeatureDetectorConfig _config;
};

typedef std::shared_ptr<FeatureDetector> FeatureDetectorPtr;

}  // namespace GSLAM
#endif  // GSLAM_AMD_FEATUREDETECTOR_H_
