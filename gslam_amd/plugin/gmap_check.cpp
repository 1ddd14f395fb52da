// gmap_check — test harness: reads a .gmap with the reference's OWN loader (GSLAM/plugins/gmap/MapHash.cpp:363-445,
// compiled from where it lies) and prints what it holds, one record per line, for tests/test_datasets_gpu.py:
//   map frames <n> points <m>
//   point <id> <x> <y> <z>
//   frame <id> <timestamp> <qx qy qz qw tx ty tz s> kps <n> obs <n> desc <rows>x<cols> first_kp <x> <y> <octave>
//   obs <frame id> {<point id> <keypoint index> <u> <v>} ...   (all observations of the frame)
//   deschash <frame id> <31-polynomial hash of the descriptor bytes>
#include <cstdio>

#include "MapHash.h"

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  MapHash map;
  if (!map.load(argv[1])) {
    fprintf(stderr, "MapHash::load(%s) failed\n", argv[1]);
    return 2;
  }
  GSLAM::FrameArray frames;
  GSLAM::PointArray points;
  map.getFrames(frames);
  map.getPoints(points);
  printf("map frames %zu points %zu\n", frames.size(), points.size());
  for (auto& p : points) {
    const GSLAM::Point3d x = p->getPose();
    printf("point %zu %.17g %.17g %.17g\n", (size_t)p->id(), x.x, x.y, x.z);
  }
  for (auto& f : frames) {
    const GSLAM::SIM3 S = f->getPoseScale();
    const GSLAM::SO3 r = S.get_rotation();
    const GSLAM::Point3d t = S.get_translation();
    std::vector<GSLAM::KeyPoint> kps((size_t)f->keyPointNum());  // (the gmap frame answers per index, MapFrame.h:81-85)
    for (size_t i = 0; i < kps.size(); ++i) f->getKeyPoint((int)i, kps[i]);
    std::map<GSLAM::PointID, size_t> obs;
    f->getObservations(obs);
    const GSLAM::GImage d = f->getDescriptor(-1);
    printf("frame %zu %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g kps %zu obs %zu desc %dx%d first_kp %.9g %.9g %d\n",
           (size_t)f->id(), f->timestamp(), r.x, r.y, r.z, r.w, t.x, t.y, t.z, S.get_scale(), kps.size(), obs.size(), d.rows, d.cols,
           kps.empty() ? -1.0 : kps[0].pt.x, kps.empty() ? -1.0 : kps[0].pt.y, kps.empty() ? -1 : kps[0].octave);
    printf("obs %zu", (size_t)f->id());
    for (auto& o : obs) printf(" %zu %zu %.9g %.9g", (size_t)o.first, o.second, kps[o.second].pt.x, kps[o.second].pt.y);
    printf("\n");
    if (d.rows > 0) {
      unsigned sum = 0;
      for (int i = 0; i < d.rows * 32; ++i) sum = sum * 31u + d.data[i];
      printf("deschash %zu %u\n", (size_t)f->id(), sum);
    }
  }
  return 0;
}
