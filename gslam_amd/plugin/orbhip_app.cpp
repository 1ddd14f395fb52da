// libgslam_orbhip.so — a GSLAM *application* plugin (GSLAM_REGISTER_APPLICATION, GSLAM/core/GSLAM.h:26-33) that puts
// the MI355X front end on GSLAM's own message bus, in the place the external ORBSLAM plugin occupies
// (doc/doxygen/4_1_orbslam.dox:10-29):
//     in   "dataset/frame"     FramePtr  (what `gslam play` publishes, plugins/play/main.cpp:16,132)
//     out  "orbhip/curframe"   FramePtr  (what qviz / metric_time / metric_traj subscribe to)
//          "orbhip/matches"    Svar {id, keypoints, matches}
// Per frame: gray image -> FeatureDetector::detectAndCompute -> MapFrame::setKeyPoints (Map.h:311-312) -> brute-force
// match against the previous frame (cross-checked, Hamming <= matchMaxDistance) -> publish.
// This is plumbing (SURVEY.md 8 f4 / BASELINE configs[0]): tracking, mapping and loop closing stay out of scope.
//     gslam play orbhip -dataset X -FeatureDetectorPlugin /path/libgslam_featuredetector.so
#include <GSLAM/core/GSLAM.h>

#include "FeatureDetector.h"

using namespace GSLAM;

int run_orbhip(Svar config) {
  svar = config;  // alias the host's registry, as every GSLAM application does
  const int n_features = config.arg<int>("orbhip.nFeatures", 1000, "ORB keypoints per frame");
  const int queue = config.arg<int>("orbhip.queue", 0, "subscriber queue (0 = handle in the publisher's thread)");
  if (config.get("help", false)) return config.help();

  FeatureDetectorPtr det = FeatureDetector::create();
  if (!det) {
    LOG(ERROR) << "orbhip: cannot load the FeatureDetector plugin (svar FeatureDetectorPlugin)";
    return -1;
  }
  det->_config.nFeatures = n_features;
  det->_config.matchCrossCheck = true;

  Publisher pub_frame = messenger.advertise<MapFrame>("orbhip/curframe", 0);
  Publisher pub_match = messenger.advertise<Svar>("orbhip/matches", 0);
  GImage last_desc;

  Subscriber sub = messenger.subscribe("dataset/frame", queue, [&](FramePtr fr) {
    if (!fr || !fr->cameraNum()) return;
    GImage img = fr->getImage(0, IMAGE_GRAY);
    if (img.empty()) img = fr->getImage(0);
    std::vector<KeyPoint> kps;
    GImage desc;
    if (!det->detectAndCompute(img, kps, desc)) {
      LOG(ERROR) << "orbhip: extraction failed on frame " << fr->id();
      return;
    }
    fr->setKeyPoints(kps, desc);
    std::vector<std::pair<int, int> > matches;
    if (!last_desc.empty() && desc.rows > 0) det->match(desc, last_desc, matches);
    last_desc = desc.clone();
    pub_match.publish(Svar({{"id", (int)fr->id()}, {"keypoints", (int)kps.size()}, {"matches", (int)matches.size()}}));
    pub_frame.publish(fr);
  });

  LOG(INFO) << "orbhip ready.";
  return Messenger::exec();
}

GSLAM_REGISTER_APPLICATION(orbhip, run_orbhip);
