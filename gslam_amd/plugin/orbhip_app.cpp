// libgslam_orbhip.so — a GSLAM *application* plugin (GSLAM_REGISTER_APPLICATION, GSLAM/core/GSLAM.h:26-33) that puts
// the MI355X hot path on GSLAM's own message bus, in the place the external ORBSLAM plugin occupies
// (doc/doxygen/4_1_orbslam.dox:10-29):
//     in   "dataset/frame"     FramePtr  (what `gslam play` publishes, plugins/play/main.cpp:16,132)
//          "dataset/status"    int       (5 = FINISHED, plugins/play/main.cpp:5-7)
//     out  "orbhip/curframe"   FramePtr  (what qviz / metric_time / metric_traj subscribe to)
//          "orbhip/map"        MapPtr    (frames + map points after every windowed bundle adjustment)
//          "orbhip/matches"    Svar {id, keypoints, matches, tracked}
// Per frame (BASELINE configs[0] "C1": extract + match to previous + optimizePnP):
//     gray image -> FeatureDetector::detectAndCompute -> MapFrame::setKeyPoints (Map.h:311-312)
//     -> brute-force match against the previous frame (cross-checked, Hamming <= matchMaxDistance)
//     -> Optimizer::optimizePnP on the matched (map point, CameraAnchor) pairs, start = previous pose (Optimizer.h:202-207)
//     -> new keypoints get map points by intersecting their rays with the ground plane z = 0 (the `synthplane` dataset's
//        scene; a monocular front end needs SOME initialisation and this one is exact for that scene)
//     every `orbhip.ba_every` frames: Optimizer::optimize on the last `orbhip.ba_window` frames (Optimizer.h:229).
// With `orbhip.vocabulary <file.gbow>` (GSLAM's own vocabulary format, core/Vocabulary.h:1843-1932) the application also
// does the place-recognition half of the ORBSLAM plugin's front end on the GPU:
//     descriptors -> Vocabulary::transform (libgslam_vocabulary: gh_bow_transform_host) -> BowVector + FeatureVector, kept in
//        an OrbhipFrame (MapFrame::getBoWVector / getFeatureVector, Map.h:322-325) that is what gets published and mapped,
//     OrbhipLoopDetector (LoopDetector, Map.h:382-395): every frame is inserted, candidates = the frames at least
//        `orbhip.loop_gap` frames old, scored in ONE batched call (scoreVocabularyBatch -> gh_bow_score_host), best first,
//     a candidate above `orbhip.loop_score` is brute-force matched; matches whose two features lie in the same vocabulary
//        node (the FeatureVector test of ORB-SLAM's SearchByBoW) are kept, and with at least `orbhip.loop_matches` of them
//        a FrameConnection (Map.h:246-262) is added to both frames and {id, candidate, score, matches} goes out on "orbhip/loop".
//     Consecutive tracked frames are linked by FrameConnections too (matches + child-to-parent SE3).
// It is a minimal tracking front end, not a SLAM system: no relocalisation, no loop CORRECTION, no map management.
//     gslam play -dataset seq.synthplane -autostart 1 orbhip metric_time -slam orbhip
// `orbhip.log <file>` records every input and output of the three plugin calls so that tests can replay them through
// the CPU checker.
#include <GSLAM/core/GSLAM.h>
#include <GSLAM/core/Optimizer.h>

#include <atomic>
#include <deque>
#include <fstream>
#include <map>
#include <thread>

#include <GSLAM/core/Vocabulary.h>

#include "FeatureDetector.h"

using namespace GSLAM;

namespace {

// FrameConnection (Map.h:246-262) with storage: the matches of a frame pair and, when known, the child-to-parent motion.
class OrbhipConnection : public FrameConnection {
 public:
  std::string type() const override { return "OrbhipConnection"; }
  int matchesNum() override { return (int)matches_.size(); }
  bool getMatches(std::vector<std::pair<int, int> >& m) override { m = matches_; return true; }
  bool getChild2Parent(SE3& T) override { if (has_pose_) T = c2p_; return has_pose_; }
  bool getChild2Parent(SIM3& S) override { if (has_pose_) S = SIM3(c2p_, 1.0); return has_pose_; }
  bool setMatches(std::vector<std::pair<int, int> >& m) override { matches_ = m; return true; }
  bool setChild2Parent(SE3& T) override { c2p_ = T; has_pose_ = true; return true; }
  bool setChild2Parent(SIM3& S) override { c2p_ = S.get_se3(); has_pose_ = true; return true; }

 private:
  std::vector<std::pair<int, int> > matches_;
  SE3 c2p_;
  bool has_pose_ = false;
};

// The frame the application publishes and maps when a vocabulary is configured: the dataset's frame (image, camera) plus
// what the front end computed -- keypoints, descriptors, BoW / feature vectors, connections.
class OrbhipFrame : public MapFrame {
 public:
  explicit OrbhipFrame(const FramePtr& src) : MapFrame(src->id(), src->timestamp()), src_(src) { setPose(src->getPoseScale()); }
  std::string type() const override { return "OrbhipFrame"; }
  int cameraNum() const override { return src_->cameraNum(); }
  SE3 getCameraPose(int idx = 0) const override { return src_->getCameraPose(idx); }
  int imageChannels(int idx = 0) const override { return src_->imageChannels(idx); }
  Camera getCamera(int idx = 0) override { return src_->getCamera(idx); }
  GImage getImage(int idx = 0, int mask = IMAGE_UNDEFINED) override { return src_->getImage(idx, mask); }
  int keyPointNum() const override { ReadMutex l(mu_); return (int)kps_.size(); }
  bool setKeyPoints(const std::vector<KeyPoint>& k, const GImage& d) override {
    WriteMutex l(mu_);
    kps_ = k;
    desc_ = d.clone();
    return true;
  }
  bool getKeyPoints(std::vector<KeyPoint>& k) const override { ReadMutex l(mu_); k = kps_; return true; }
  bool getKeyPoint(int idx, KeyPoint& pt) const override {
    ReadMutex l(mu_);
    if (idx < 0 || idx >= (int)kps_.size()) return false;
    pt = kps_[idx];
    return true;
  }
  bool getKeyPoint(int idx, Point2f& pt) const override {
    ReadMutex l(mu_);
    if (idx < 0 || idx >= (int)kps_.size()) return false;
    pt = kps_[idx].pt;
    return true;
  }
  GImage getDescriptor(int idx = -1) const override {
    ReadMutex l(mu_);
    if (idx < 0) return desc_;
    return idx < desc_.rows ? desc_.row(idx) : GImage();
  }
  bool getBoWVector(BowVector& v) const override { ReadMutex l(mu_); v = bow_; return !bow_.empty(); }
  bool getFeatureVector(FeatureVector& v) const override { ReadMutex l(mu_); v = feat_; return !feat_.empty(); }
  void setBoW(const BowVector& b, const FeatureVector& f) { WriteMutex l(mu_); bow_ = b; feat_ = f; }
  FrameConnectionPtr getParent(FrameID id) const override { ReadMutex l(mu_); auto it = parents_.find(id); return it == parents_.end() ? FrameConnectionPtr() : it->second; }
  FrameConnectionPtr getChild(FrameID id) const override { ReadMutex l(mu_); auto it = children_.find(id); return it == children_.end() ? FrameConnectionPtr() : it->second; }
  bool getParents(FrameConnectionMap& p) const override { ReadMutex l(mu_); p = parents_; return true; }
  bool getChildren(FrameConnectionMap& c) const override { ReadMutex l(mu_); c = children_; return true; }
  bool addParent(FrameID id, const FrameConnectionPtr& c) override { WriteMutex l(mu_); parents_[id] = c; return true; }
  bool addChildren(FrameID id, const FrameConnectionPtr& c) override { WriteMutex l(mu_); children_[id] = c; return true; }
  bool eraseParent(FrameID id) override { WriteMutex l(mu_); return parents_.erase(id) > 0; }
  bool eraseChild(FrameID id) override { WriteMutex l(mu_); return children_.erase(id) > 0; }
  bool clearParents() override { WriteMutex l(mu_); parents_.clear(); return true; }
  bool clearChildren() override { WriteMutex l(mu_); children_.clear(); return true; }

 private:
  FramePtr src_;
  mutable MutexRW mu_;
  std::vector<KeyPoint> kps_;
  GImage desc_;
  BowVector bow_;
  FeatureVector feat_;
  FrameConnectionMap parents_, children_;
};

// LoopDetector (Map.h:382-395) on BoW vectors: all candidates older than `gap` frames are scored in one batched GPU call.
class OrbhipLoopDetector : public LoopDetector {
 public:
  typedef bool (*score_fn)(const Vocabulary*, const BowVector*, const BowVector* const*, int, double*);
  OrbhipLoopDetector(const std::shared_ptr<Vocabulary>& voc, score_fn score, int gap, double min_score)
      : voc_(voc), score_(score), gap_(gap), min_score_(min_score) {}
  std::string type() const override { return "OrbhipLoopDetector"; }
  bool insertMapFrame(const FramePtr& f) override {
    BowVector v;
    if (!f || !f->getBoWVector(v)) return false;
    entries_.push_back(std::make_pair(f->id(), v));
    return true;
  }
  bool eraseMapFrame(const FrameID& id) override {
    for (size_t i = 0; i < entries_.size(); ++i)
      if (entries_[i].first == id) {
        entries_.erase(entries_.begin() + (long)i);
        return true;
      }
    return false;
  }
  bool obtainCandidates(const FramePtr& f, LoopCandidates& out) override {
    out.clear();
    BowVector q;
    if (!f || !f->getBoWVector(q) || !score_) return false;
    std::vector<const BowVector*> db;
    std::vector<FrameID> ids;
    for (auto& e : entries_)
      if (e.first + (FrameID)gap_ <= f->id()) {  // old enough not to be a neighbour of the query
        db.push_back(&e.second);
        ids.push_back(e.first);
      }
    if (db.empty()) return true;
    std::vector<double> sc(db.size(), 0.0);
    if (!score_(voc_.get(), &q, db.data(), (int)db.size(), sc.data())) return false;
    for (size_t i = 0; i < db.size(); ++i)
      if (sc[i] >= min_score_) out.push_back(LoopCandidate(ids[i], sc[i]));
    std::stable_sort(out.begin(), out.end(), [](const LoopCandidate& a, const LoopCandidate& b) { return a.score > b.score; });
    return true;
  }

 private:
  std::shared_ptr<Vocabulary> voc_;
  score_fn score_;
  int gap_;
  double min_score_;
  std::vector<std::pair<FrameID, BowVector> > entries_;
};

class OrbhipPoint : public MapPoint {
 public:
  OrbhipPoint(PointID id, const Point3d& p) : MapPoint(id, p) {}
  std::string type() const override { return "OrbhipPoint"; }
};

class OrbhipMap : public Map {
 public:
  std::string type() const override { return "OrbhipMap"; }
  bool insertMapPoint(const PointPtr& p) override { WriteMutex l(mu_); points_[p->id()] = p; return true; }
  bool insertMapFrame(const FramePtr& f) override { WriteMutex l(mu_); frames_[f->id()] = f; return true; }
  std::size_t frameNum() const override { ReadMutex l(mu_); return frames_.size(); }
  std::size_t pointNum() const override { ReadMutex l(mu_); return points_.size(); }
  FramePtr getFrame(const FrameID& id) const override {
    ReadMutex l(mu_);
    auto it = frames_.find(id);
    return it == frames_.end() ? FramePtr() : it->second;
  }
  PointPtr getPoint(const PointID& id) const override {
    ReadMutex l(mu_);
    auto it = points_.find(id);
    return it == points_.end() ? PointPtr() : it->second;
  }
  bool getFrames(FrameArray& frames) const override {
    ReadMutex l(mu_);
    for (auto& kv : frames_) frames.push_back(kv.second);
    return true;
  }
  bool getPoints(PointArray& points) const override {
    ReadMutex l(mu_);
    for (auto& kv : points_) points.push_back(kv.second);
    return true;
  }

  // what the front end knows about a frame beyond the frame object itself: keypoints, descriptors and which map point
  // each keypoint observes (dataset frame classes such as the reference's FrameMono keep none of it)
  void setFeatures(FrameID id, const std::vector<KeyPoint>& kps, const GImage& desc,
                   const std::vector<std::pair<PointID, size_t> >& obs) {
    WriteMutex l(mu_);
    FrameData& d = features_[id];
    d.kps = kps;
    d.desc = desc.clone();
    d.obs = obs;
  }

  // The reference's map file (`.gmap`: "Hash" / "binary", GSLAM/plugins/gmap/MapHash.cpp:278-360 writes it, :363-445 reads
  // it): so that `gslam orbhip gmap play ... -map orbhip/map -out map.gmap` (the reference's own gmap application calls
  // Map::save on whatever map is published, plugins/gmap/main.cpp:11-13) leaves a file GSLAM's MapHash::load, its gmap
  // viewer and its evaluation tools read.  Field order and raw-struct encoding are the reference's OutStream
  // (:207-236: every value as its in-memory bytes, vectors as size_t count + elements, GImage as cols rows flags + data,
  // strings as size_t length + bytes).  Unlike the reference (which writes empty images there) the descriptors are kept.
  bool save(std::string path) const override {
    if (path.empty() || path.find(".gmap") == std::string::npos) return false;
    std::ofstream ofs(path.c_str(), std::ios::out | std::ios::binary);
    if (!ofs.is_open()) return false;
    ReadMutex l(mu_);
    auto raw = [&ofs](const void* p, size_t n) { ofs.write((const char*)p, (std::streamsize)n); };
    auto put_image = [&](const GImage& im) {
      const int hdr[3] = {im.cols, im.rows, im.flags};
      raw(hdr, sizeof(hdr));
      if (!im.empty()) raw(im.data, (size_t)im.total() * im.elemSize());
    };
    auto put_doubles = [&](const std::vector<double>& v) {
      const size_t n = v.size();
      raw(&n, sizeof(n));
      if (n) raw(v.data(), n * sizeof(double));
    };
    ofs << "Hash" << std::endl << "binary" << std::endl;
    const size_t nf = frames_.size(), np = points_.size();
    raw(&nf, sizeof(nf));
    raw(&np, sizeof(np));
    for (auto& kv : points_) {
      const PointPtr& pt = kv.second;
      const PointID id = pt->id();
      const Point3d pos = pt->getPose(), nrm = pt->getNormal();
      const ColorType col = pt->getColor();
      const FrameID ref = pt->refKeyframeID();
      raw(&id, sizeof(id));
      raw(&pos, sizeof(pos));
      raw(&nrm, sizeof(nrm));
      raw(&col, sizeof(col));
      raw(&ref, sizeof(ref));
      put_image(GImage());
    }
    for (auto& kv : frames_) {
      const FramePtr& fr = kv.second;
      auto fit = features_.find(kv.first);
      static const FrameData none;
      const FrameData& fd = fit == features_.end() ? none : fit->second;
      const FrameID id = fr->id();
      const double stamp = fr->timestamp();
      const SIM3 pose = fr->getPoseScale();
      raw(&id, sizeof(id));
      raw(&stamp, sizeof(stamp));
      raw(&pose, sizeof(pose));
      put_image(GImage());  // the image itself stays with the dataset
      const std::string img_file;
      const size_t slen = img_file.size();
      raw(&slen, sizeof(slen));
      const int channels = fr->imageChannels(0);
      raw(&channels, sizeof(channels));
      put_doubles(fr->getCamera(0).getParameters());
      put_doubles(std::vector<double>());  // no GPS
      put_image(fd.desc);
      const size_t nk = fd.kps.size();
      raw(&nk, sizeof(nk));
      if (nk) raw(fd.kps.data(), nk * sizeof(KeyPoint));
      raw(&nk, sizeof(nk));  // one colour per keypoint (MapHash::load asserts the sizes agree)
      for (size_t i = 0; i < nk; ++i) {
        const ColorType white(255, 255, 255);
        raw(&white, sizeof(white));
      }
      // only observations of points that are in the map (a point enters it with its first bundle adjustment)
      std::vector<std::pair<PointID, size_t> > obs;
      for (size_t i = 0; i < fd.obs.size(); ++i)
        if (points_.count(fd.obs[i].first)) obs.push_back(fd.obs[i]);
      const size_t no = obs.size();
      raw(&no, sizeof(no));
      for (size_t i = 0; i < no; ++i) raw(&obs[i], sizeof(obs[i]));
      const size_t zero = 0;
      raw(&zero, sizeof(zero));  // children
      raw(&zero, sizeof(zero));  // parents
    }
    return ofs.good();
  }

 private:
  struct FrameData {
    std::vector<KeyPoint> kps;
    GImage desc;
    std::vector<std::pair<PointID, size_t> > obs;
  };
  mutable MutexRW mu_;
  std::map<FrameID, FramePtr> frames_;
  std::map<PointID, PointPtr> points_;
  std::map<FrameID, FrameData> features_;
};

struct TrackedFrame {
  FramePtr frame;
  SE3 pose;                      // T_wc
  std::vector<KeyPoint> kps;
  std::vector<Point2d> anchors;  // camera.UnProject(kp.pt).xy (z = 1 plane)
  std::vector<int64_t> pid;      // map point id per keypoint, -1 = none
};

template <typename T>
void put(std::ofstream& o, const T& v) { o.write((const char*)&v, sizeof(T)); }
void put_pose(std::ofstream& o, const SE3& T) {
  const SO3 r = T.get_rotation();
  const Point3d t = T.get_translation();
  const double p[7] = {r.x, r.y, r.z, r.w, t.x, t.y, t.z};
  o.write((const char*)p, sizeof(p));
}

}  // namespace

int run_orbhip(Svar config) {
  svar = config;  // alias the host's registry, as every GSLAM application does
  const int n_features = config.arg<int>("orbhip.nFeatures", 1000, "ORB keypoints per frame");
  const int queue = config.arg<int>("orbhip.queue", 0, "subscriber queue (0 = handle in the publisher's thread)");
  const int ba_every = config.arg<int>("orbhip.ba_every", 10, "windowed bundle adjustment every N frames (0 = never)");
  const int ba_window = config.arg<int>("orbhip.ba_window", 10, "frames in the bundle-adjustment window");
  const int min_track = config.arg<int>("orbhip.min_track", 30, "minimum 3D-2D matches for optimizePnP");
  const double inlier = config.arg<double>("orbhip.inlier", 0.006, "inlier radius of the PnP refit, normalised image units");
  const bool track = config.arg<bool>("orbhip.track", true, "run optimizePnP / optimize (false: extract + match only)");
  const bool stop_on_finish = config.arg<bool>("orbhip.stop_on_finish", false, "publish messenger/stop when the dataset ends");
  const bool start_dataset = config.arg<bool>("orbhip.start_dataset", false,
                                              "publish qviz/start (what the GUI's play button does) until the first frame "
                                              "arrives, so that no frame is lost while the plugins load");
  const std::string log_path = config.arg<std::string>("orbhip.log", "", "binary record of every plugin call (tests)");
  const std::string voc_path = config.arg<std::string>("orbhip.vocabulary", "", ".gbow vocabulary: BoW vectors per frame, loop candidates");
  const int loop_gap = config.arg<int>("orbhip.loop_gap", 20, "frames between a query and its oldest-allowed loop candidate");
  const double loop_score = config.arg<double>("orbhip.loop_score", 0.05, "minimum BoW score of a loop candidate");
  const int loop_matches = config.arg<int>("orbhip.loop_matches", 40, "node-consistent matches that make a loop connection");
  const int levels_up = config.arg<int>("orbhip.levels_up", 4, "FeatureVector level (levels up from the leaves)");
  if (config.get("help", false)) return config.help();

  FeatureDetectorPtr det = FeatureDetector::create();
  if (!det) {
    LOG(ERROR) << "orbhip: cannot load the FeatureDetector plugin (svar FeatureDetectorPlugin)";
    return -1;
  }
  det->_config.nFeatures = n_features;
  det->_config.matchCrossCheck = true;
  OptimizerPtr opt;
  if (track) {
    opt = Optimizer::create();
    if (!opt) LOG(WARNING) << "orbhip: no Optimizer plugin (svar OptimizerPlugin): tracking disabled";
  }
  // `.gmap` file (the reference's map format, OrbhipMap::save) rewritten after every bundle adjustment.  The reference's
  // own gmap application would do the same for any published map (plugins/gmap/main.cpp:11-13: Map::save on the "map"
  // topic), but in this snapshot loading it next to `play` makes play's "qviz/open" / frame callbacks fire two and three
  // times (reproduced with reference plugins only), so the application saves its map itself.
  const std::string save_map = config.arg<std::string>("orbhip.save_map", "", "write the map as a .gmap file after every BA");
  if (opt) opt->_config.maxIterations = config.arg<int>("orbhip.max_iterations", 30, "LM iterations per call");

  // vocabulary plugin (libgslam_vocabulary.so: the subclass of GSLAM::Vocabulary whose transforms and batched scoring run on the GPU)
  std::shared_ptr<Vocabulary> voc;
  std::shared_ptr<OrbhipLoopDetector> loops;
  if (!voc_path.empty()) {
    typedef std::shared_ptr<Vocabulary> (*factory_t)(const char*);
    SharedLibraryPtr lib = Registry::get(svar.GetString("VocabularyPlugin", "libgslam_vocabulary"));
    factory_t f = lib ? (factory_t)lib->getSymbol("createVocabularyInstance") : NULL;
    OrbhipLoopDetector::score_fn sf = lib ? (OrbhipLoopDetector::score_fn)lib->getSymbol("scoreVocabularyBatch") : NULL;
    if (f) voc = f(voc_path.c_str());
    if (!voc || !sf) {
      LOG(ERROR) << "orbhip: cannot load vocabulary " << voc_path << " through the Vocabulary plugin (svar VocabularyPlugin)";
      return -1;
    }
    loops.reset(new OrbhipLoopDetector(voc, sf, loop_gap, loop_score));
  }
  std::map<FrameID, std::shared_ptr<OrbhipFrame> > bow_frames;  // frames by id (loop candidates are looked up here)
  Publisher pub_loop = messenger.advertise<Svar>("orbhip/loop", 0);
  Publisher pub_frame = messenger.advertise<MapFrame>("orbhip/curframe", 0);
  Publisher pub_map = messenger.advertise<Map>("orbhip/map", 0);
  Publisher pub_match = messenger.advertise<Svar>("orbhip/matches", 0);
  std::shared_ptr<OrbhipMap> map(new OrbhipMap());
  std::map<int64_t, Point3d> points;  // map point id -> world position
  std::deque<TrackedFrame> window;
  GImage last_desc;
  int64_t next_pid = 1;
  int n_frames = 0;
  std::ofstream log;
  if (!log_path.empty()) log.open(log_path.c_str(), std::ios::binary);

  // ray of a keypoint through the plane z = 0
  auto on_plane = [](const SE3& Twc, const Point2d& a, Point3d& X) {
    const Point3d d = Twc.get_rotation() * Point3d(a.x, a.y, 1.0), c = Twc.get_translation();
    if (!(fabs(d.z) > 1e-9)) return false;
    const double lam = -c.z / d.z;
    if (!(lam > 0)) return false;
    X = c + d * lam;
    return true;
  };

  Subscriber sub = messenger.subscribe("dataset/frame", queue, [&](FramePtr fr) {
    if (!fr || !fr->cameraNum()) return;
    std::shared_ptr<OrbhipFrame> of;
    if (voc) {  // the frame that is published and mapped carries the BoW data: the dataset's frame classes cannot
      of.reset(new OrbhipFrame(fr));
      fr = of;
    }
    GImage img = fr->getImage(0, IMAGE_GRAY);
    if (img.empty()) img = fr->getImage(0);
    TrackedFrame cur;
    cur.frame = fr;
    GImage desc;
    if (!det->detectAndCompute(img, cur.kps, desc)) {
      LOG(ERROR) << "orbhip: extraction failed on frame " << fr->id();
      return;
    }
    fr->setKeyPoints(cur.kps, desc);
    std::vector<std::pair<int, int> > matches;
    if (!last_desc.empty() && desc.rows > 0) det->match(desc, last_desc, matches);
    if (of && desc.rows > 0) {
      BowVector bow;
      FeatureVector feat;
      voc->transform(desc, bow, feat, levels_up);
      of->setBoW(bow, feat);
      if (log.is_open()) {
        put(log, (int32_t)4);  // record type 4: BoW vector + feature vector of the frame
        put(log, (int32_t)fr->id());
        put(log, (int32_t)bow.size());
        for (auto& kv : bow) { put(log, (uint32_t)kv.first); put(log, (float)kv.second); }
        put(log, (int32_t)feat.size());
        for (auto& kv : feat) {
          put(log, (uint32_t)kv.first);
          put(log, (int32_t)kv.second.size());
          for (unsigned int i : kv.second) put(log, (uint32_t)i);
        }
      }
      // loop candidates: one batched scoring call over every frame that is old enough, then the best one is verified
      LoopCandidates cands;
      loops->obtainCandidates(fr, cands);
      if (log.is_open()) {
        put(log, (int32_t)5);  // record type 5: loop candidates (best first) and the verified connection, if any
        put(log, (int32_t)fr->id());
        put(log, (int32_t)cands.size());
        for (auto& c : cands) { put(log, (int32_t)c.frameId); put(log, (double)c.score); }
      }
      int32_t loop_to = -1;
      std::vector<std::pair<int, int> > lm;
      if (!cands.empty()) {
        std::shared_ptr<OrbhipFrame> old = bow_frames[cands[0].frameId];
        std::vector<std::pair<int, int> > raw;
        if (old && det->match(desc, old->getDescriptor(), raw)) {
          // the FeatureVector test of ORB-SLAM's SearchByBoW: both features descend to the same node `levels_up` above the leaves
          FeatureVector fo;
          old->getFeatureVector(fo);
          std::vector<uint32_t> node_q((size_t)desc.rows, 0u), node_t((size_t)old->keyPointNum(), 0u);
          for (auto& kv : feat)
            for (unsigned int i : kv.second) node_q[i] = (uint32_t)kv.first + 1u;
          for (auto& kv : fo)
            for (unsigned int i : kv.second) node_t[i] = (uint32_t)kv.first + 1u;
          for (auto& m : raw)
            if (node_q[m.first] != 0u && node_q[m.first] == node_t[m.second]) lm.push_back(m);
          if ((int)lm.size() >= loop_matches) {
            loop_to = (int32_t)old->id();
            FrameConnectionPtr c(new OrbhipConnection());
            c->setMatches(lm);
            of->addParent(old->id(), c);
            old->addChildren(of->id(), c);
            pub_loop.publish(Svar({{"id", (int)fr->id()}, {"candidate", (int)old->id()}, {"score", cands[0].score}, {"matches", (int)lm.size()}}));
          }
        }
      }
      if (log.is_open()) {
        put(log, loop_to);
        put(log, (int32_t)lm.size());
        for (auto& m : lm) { put(log, (int32_t)m.first); put(log, (int32_t)m.second); }
      }
      loops->insertMapFrame(fr);
      bow_frames[fr->id()] = of;
    }
    const Camera cam = fr->getCamera(0);
    const int n = (int)cur.kps.size();
    cur.anchors.resize(n);
    cur.pid.assign(n, -1);
    for (int i = 0; i < n; ++i) {
      const Point3d a = cam.isValid() ? cam.UnProject(Point2d(cur.kps[i].pt.x, cur.kps[i].pt.y)) : Point3d(0, 0, 1);
      cur.anchors[i] = Point2d(a.x / a.z, a.y / a.z);
    }
    if (log.is_open()) {
      put(log, (int32_t)1);  // record type 1: frame
      put(log, (int32_t)fr->id());
      put(log, (int32_t)n);
      if (n) log.write((const char*)cur.kps.data(), (std::streamsize)n * sizeof(KeyPoint));
      if (n) log.write((const char*)desc.data, (std::streamsize)n * 32);
      put(log, (int32_t)matches.size());
      for (auto& m : matches) { put(log, (int32_t)m.first); put(log, (int32_t)m.second); }
    }

    int tracked = 0;
    bool have_pose = false;
    if (opt && cam.isValid()) {
      if (window.empty()) {
        cur.pose = fr->getPose();  // the gauge: the dataset's pose of the first frame
        have_pose = true;
      } else {
        const TrackedFrame& prev = window.back();
        std::vector<std::pair<Point3d, CameraAnchor> > m3d;
        std::vector<std::pair<int, int64_t> > who;
        for (auto& m : matches) {
          const int64_t id = prev.pid[m.second];
          if (id < 0) continue;
          m3d.push_back(std::make_pair(points[id], CameraAnchor(cur.anchors[m.first].x, cur.anchors[m.first].y, 1.0)));
          who.push_back(std::make_pair(m.first, id));
        }
        tracked = (int)m3d.size();
        if (tracked >= min_track) {
          SE3 pose = prev.pose;
          bool ok = true;
          // two rounds, as ORB-SLAM's pose optimisation does: Huber-robust fit on every match, then a refit on the
          // matches within `inlier` of the first fit (cross-checked Hamming matches still hold ~7 % wrong pairs on this
          // texture, and a match kept here hands its map point on to the new frame)
          for (int round = 0; round < 2 && ok; ++round) {
            const SE3 start = pose;
            ok = opt->optimizePnP(m3d, pose, UPDATE_KF_SE3, NULL);
            if (log.is_open()) {
              put(log, (int32_t)2);  // record type 2: optimizePnP call
              put(log, (int32_t)fr->id());
              put(log, (int32_t)m3d.size());
              for (auto& p : m3d) {
                const double r[5] = {p.first.x, p.first.y, p.first.z, p.second.x, p.second.y};
                log.write((const char*)r, sizeof(r));
              }
              put_pose(log, start);
              put_pose(log, pose);
              put(log, (int32_t)(ok ? 1 : 0));
            }
            if (!ok || round == 1) break;
            std::vector<std::pair<Point3d, CameraAnchor> > in3d;
            std::vector<std::pair<int, int64_t> > inwho;
            const SE3 Tcw = pose.inverse();
            for (size_t k = 0; k < m3d.size(); ++k) {
              const Point3d Xc = Tcw * m3d[k].first;
              if (!(Xc.z > 1e-9)) continue;
              const double dx = Xc.x / Xc.z - m3d[k].second.x, dy = Xc.y / Xc.z - m3d[k].second.y;
              if (dx * dx + dy * dy < inlier * inlier) {
                in3d.push_back(m3d[k]);
                inwho.push_back(who[k]);
              }
            }
            if ((int)in3d.size() < min_track) break;  // keep the first fit
            m3d.swap(in3d);
            who.swap(inwho);
          }
          tracked = (int)m3d.size();
          if (ok) {
            cur.pose = pose;
            have_pose = true;
            for (auto& w : who) cur.pid[w.first] = w.second;
          }
        }
      }
      if (have_pose) {
        fr->setPose(cur.pose);
        for (int i = 0; i < n; ++i) {  // new map points for keypoints that are not tracked yet
          if (cur.pid[i] >= 0) continue;
          Point3d X;
          if (!on_plane(cur.pose, cur.anchors[i], X)) continue;
          cur.pid[i] = next_pid;
          points[next_pid++] = X;
        }
        if (of && !window.empty()) {  // FrameConnection child (this frame) -> parent (the previous tracked frame)
          std::shared_ptr<OrbhipFrame> pf = std::dynamic_pointer_cast<OrbhipFrame>(window.back().frame);
          if (pf) {
            FrameConnectionPtr c(new OrbhipConnection());
            c->setMatches(matches);
            SE3 c2p = window.back().pose.inverse() * cur.pose;
            c->setChild2Parent(c2p);
            of->addParent(pf->id(), c);
            pf->addChildren(of->id(), c);
          }
        }
        window.push_back(cur);
        while ((int)window.size() > ba_window) window.pop_front();
        map->insertMapFrame(fr);
        {
          std::vector<std::pair<PointID, size_t> > obs;
          for (int i = 0; i < n; ++i)
            if (cur.pid[i] >= 0) obs.push_back(std::make_pair((PointID)cur.pid[i], (size_t)i));
          map->setFeatures(fr->id(), cur.kps, desc, obs);
        }
      } else {
        window.clear();  // lost: start again from the next frame's dataset pose
      }
    }
    last_desc = desc.clone();
    ++n_frames;

    // windowed bundle adjustment over the frames in the window (the two oldest fixed: the gauge)
    if (opt && have_pose && ba_every > 0 && n_frames % ba_every == 0 && window.size() >= 3) {
      BundleGraph g;
      g.cameraDOF = UPDATE_CAMERA_NONE;
      std::map<int64_t, int> count;
      for (auto& f : window)
        for (int64_t id : f.pid)
          if (id >= 0) ++count[id];
      std::map<int64_t, size_t> slot;
      for (auto& kv : count)
        if (kv.second >= 2) {
          slot[kv.first] = g.mappoints.size();
          g.mappoints.push_back(std::make_pair(points[kv.first], true));
        }
      for (size_t fi = 0; fi < window.size(); ++fi) {
        KeyFrameEstimzation kf;
        kf.estimation = SIM3(window[fi].pose, 1.0);
        kf.dof = fi < 2 ? UPDATE_KF_NONE : UPDATE_KF_SE3;  // two fixed frames: pose AND scale gauge of a monocular window
        g.keyframes.push_back(kf);
        for (size_t i = 0; i < window[fi].pid.size(); ++i) {
          auto it = slot.find(window[fi].pid[i]);
          if (it == slot.end()) continue;
          BundleEdge e;
          e.pointId = it->second;
          e.frameId = fi;
          e.measurement = CameraAnchor(window[fi].anchors[i].x, window[fi].anchors[i].y, 1.0);
          e.information = NULL;
          g.mappointObserves.push_back(e);
        }
      }
      if (log.is_open()) {
        put(log, (int32_t)3);  // record type 3: optimize call (inputs)
        put(log, (int32_t)fr->id());
        put(log, (int32_t)g.keyframes.size());
        put(log, (int32_t)g.mappoints.size());
        put(log, (int32_t)g.mappointObserves.size());
        for (auto& kf : g.keyframes) { put_pose(log, kf.estimation.get_se3()); put(log, (int32_t)kf.dof); }
        for (auto& mp : g.mappoints) { const double p[3] = {mp.first.x, mp.first.y, mp.first.z}; log.write((const char*)p, sizeof(p)); }
        for (auto& e : g.mappointObserves) {
          put(log, (int32_t)e.frameId);
          put(log, (int32_t)e.pointId);
          const double m[2] = {e.measurement.x, e.measurement.y};
          log.write((const char*)m, sizeof(m));
        }
      }
      const bool ok = opt->optimize(g);
      if (log.is_open()) {
        put(log, (int32_t)(ok ? 1 : 0));
        for (auto& kf : g.keyframes) put_pose(log, kf.estimation.get_se3());
        for (auto& mp : g.mappoints) { const double p[3] = {mp.first.x, mp.first.y, mp.first.z}; log.write((const char*)p, sizeof(p)); }
        log.flush();
      }
      if (ok) {
        for (size_t fi = 0; fi < window.size(); ++fi) {
          window[fi].pose = g.keyframes[fi].estimation.get_se3();
          window[fi].frame->setPose(window[fi].pose);
        }
        for (auto& kv : slot) {
          points[kv.first] = g.mappoints[kv.second].first;
          map->insertMapPoint(PointPtr(new OrbhipPoint((PointID)kv.first, points[kv.first])));
        }
        pub_map.publish(std::static_pointer_cast<Map>(map));
        if (!save_map.empty() && !map->save(save_map)) LOG(ERROR) << "orbhip: cannot write " << save_map;
      }
    }
    pub_match.publish(Svar({{"id", (int)fr->id()}, {"keypoints", n}, {"matches", (int)matches.size()}, {"tracked", tracked}}));
    pub_frame.publish(fr);
  });

  Subscriber sub_status = messenger.subscribe("dataset/status", 0, [&](int status) {
    if (stop_on_finish && status == 5 && n_frames > 0) {  // FINISHED (plugins/play/main.cpp:5-7)
      if (log.is_open()) log.flush();
      messenger.publish("messenger/stop", true);
    }
  });

  LOG(INFO) << "orbhip ready.";
  std::atomic<bool> exiting(false);
  std::thread kick;
  if (start_dataset)
    kick = std::thread([&]() {
      // `play` only reacts to qviz/start once its dataset is open (plugins/play/main.cpp:30-41); repeat until frames flow
      for (int i = 0; i < 600 && !exiting && n_frames == 0; ++i) {
        messenger.publish("qviz/start", true);
        Rate::sleep(0.1);
      }
    });
  const int rc = Messenger::exec();
  exiting = true;
  if (kick.joinable()) kick.join();
  if (log.is_open()) log.close();
  return rc;
}

GSLAM_REGISTER_APPLICATION(orbhip, run_orbhip);
