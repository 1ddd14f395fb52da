// libgslam_featuredetector.so — ORB extractor + brute-force Hamming matcher plugin on the MI355X.
// Implements gslam_amd/plugin/FeatureDetector.h on top of the C ABI (gh_orb_*, gh_bf_*, gh_match_mask_dev).
// Outputs use GSLAM's own containers: std::vector<GSLAM::KeyPoint> (GSLAM/core/Map.h:122-195, layout
// identical to gh_keypoint) and an N x 32 8UC1 GImage (MapFrame::setKeyPoints, Map.h:311-312).
#include "FeatureDetector.h"

#include <algorithm>
#include <cstring>
#include <fstream>
#include <mutex>
#include <vector>

#include "gslam_hip.h"

static_assert(sizeof(GSLAM::KeyPoint) == sizeof(gh_keypoint), "KeyPoint layout must match gh_keypoint");

namespace {

class FeatureDetectorHIP : public GSLAM::FeatureDetector {
 public:
  FeatureDetectorHIP() : ctx_(nullptr), plan_(nullptr), pw_(0), ph_(0), pk_(0), pl_(0), pi_(0), pm_(0) {}
  ~FeatureDetectorHIP() override {
    if (batch_.s) gh_orb_stream_destroy(batch_.s);
    if (async_.s) gh_orb_stream_destroy(async_.s);
    if (plan_) gh_orb_plan_destroy(plan_);
    free_device_buffers();
    if (ctx_) gh_ctx_destroy(ctx_);
  }

  bool detectAndCompute(const GSLAM::GImage& image, std::vector<GSLAM::KeyPoint>& keypoints,
                        GSLAM::GImage& descriptors) override {
    std::lock_guard<std::mutex> lock(mu_);
    if (image.empty() || image.elemSize1() != 1 || !context()) return false;
    const int w = image.cols, h = image.rows, ch = image.channels();
    if (ch != 1 && ch != 3 && ch != 4) return false;
    if (!plan_ || pw_ != w || ph_ != h || pk_ != _config.nFeatures || pl_ != _config.nLevels ||
        pi_ != _config.iniThFAST || pm_ != _config.minThFAST) {
      if (plan_) gh_orb_plan_destroy(plan_);
      plan_ = nullptr;
      gh_orb_params p;
      gh_orb_default_params(&p);
      p.n_features = _config.nFeatures;
      p.n_levels = _config.nLevels;
      p.ini_th_fast = _config.iniThFAST;
      p.min_th_fast = _config.minThFAST;
      if (gh_orb_plan_create(ctx_, w, h, 1, &p, &plan_) != GH_OK) return fail("gh_orb_plan_create");
      if (!configure(plan_)) return false;
      pw_ = w; ph_ = h; pk_ = _config.nFeatures; pl_ = _config.nLevels;
      pi_ = _config.iniThFAST; pm_ = _config.minThFAST;
    }
    const int K = _config.nFeatures;
    std::vector<gh_keypoint> kps((size_t)K);
    std::vector<uint8_t> desc((size_t)K * 32);
    int32_t n = 0;
    if (ch == 1) {
      if (gh_orb_extract_host(plan_, image.data, w, kps.data(), desc.data(), &n) != GH_OK)
        return fail("gh_orb_extract_host");
    } else {
      // colour input (datasets deliver BGR/BGRA: GSLAM/plugins/datasets/IO.h:86-113): fixed-point luma on the GPU, and
      // the gray image never leaves the device: upload BGR -> luma -> extract -> one download of the result block.
      // The device buffers live as long as the geometry does (no hipMalloc / hipFree per frame).
      const size_t bgr_bytes = (size_t)w * h * ch, pitch = ((size_t)w + 63) & ~(size_t)63;
      const size_t off_kps = 256, off_desc = off_kps + (((size_t)K * sizeof(gh_keypoint) + 255) & ~(size_t)255);
      const size_t out_bytes = off_desc + (size_t)K * 32;
      if (bgr_cap_ < bgr_bytes || gray_cap_ < pitch * h + 256 || out_cap_ < out_bytes) {
        free_device_buffers();
        if (gh_dev_alloc(ctx_, bgr_bytes, &d_bgr_) != GH_OK || gh_dev_alloc(ctx_, pitch * h + 256, &d_gray_) != GH_OK ||
            gh_dev_alloc(ctx_, out_bytes, &d_out_) != GH_OK) {
          free_device_buffers();
          return fail("gh_dev_alloc");
        }
        bgr_cap_ = bgr_bytes;
        gray_cap_ = pitch * h + 256;
        out_cap_ = out_bytes;
      }
      uint8_t* out = (uint8_t*)d_out_;
      host_out_.resize(out_bytes);
      const bool ok =
          gh_dev_upload(ctx_, d_bgr_, image.data, bgr_bytes) == GH_OK &&
          gh_bgr_to_gray_dev(ctx_, (const uint8_t*)d_bgr_, w, h, ch, w * ch, (uint8_t*)d_gray_, (int)pitch) == GH_OK &&
          gh_orb_extract_dev(plan_, (const uint8_t*)d_gray_, 1, pitch * h, (int)pitch, (gh_keypoint*)(out + off_kps),
                             out + off_desc, (int32_t*)out) == GH_OK &&
          gh_dev_download(ctx_, host_out_.data(), d_out_, out_bytes) == GH_OK;
      if (!ok) return fail("bgr -> gray -> extract");
      std::memcpy(&n, host_out_.data(), sizeof(n));
      std::memcpy(kps.data(), host_out_.data() + off_kps, (size_t)K * sizeof(gh_keypoint));
      std::memcpy(desc.data(), host_out_.data() + off_desc, (size_t)K * 32);
    }
    keypoints.resize((size_t)n);
    if (n > 0) std::memcpy((void*)keypoints.data(), kps.data(), (size_t)n * sizeof(gh_keypoint));
    descriptors = GSLAM::GImage(n, 32, GSLAM::GImageType<uchar>::Type, desc.data(), true);
    return true;
  }

  // ---- batch + asynchronous entries: gh_orb_stream_* (pinned ring, three HIP streams, exact-size results) -----------
  bool detectAndComputeBatch(const std::vector<GSLAM::GImage>& images, std::vector<std::vector<GSLAM::KeyPoint> >& keypoints,
                             std::vector<GSLAM::GImage>& descriptors) override {
    std::lock_guard<std::mutex> lock(mu_);
    const size_t n = images.size();
    keypoints.assign(n, std::vector<GSLAM::KeyPoint>());
    descriptors.assign(n, GSLAM::GImage());
    if (n == 0) return true;
    if (!context()) return false;
    for (size_t i = 0; i < n; ++i)
      if (images[i].empty() || images[i].elemSize1() != 1 || images[i].cols != images[0].cols || images[i].rows != images[0].rows ||
          images[i].channels() != images[0].channels())
        return false;  // a burst is same-sized by contract
    const int chunk = (int)std::min<size_t>(n, (size_t)svar.GetInt("FeatureDetectorHIP.BatchChunk", 16));
    if (!ensure_stream(batch_, images[0], chunk, 3)) return false;
    const size_t fbytes = batch_.frame_bytes;
    std::vector<std::pair<int64_t, size_t> > inflight;  // (ticket, first image)
    auto drain_one = [&]() -> bool {
      gh_orb_stream_result r;
      if (gh_orb_stream_collect(batch_.s, inflight.front().first, &r) != GH_OK) return fail("gh_orb_stream_collect");
      const size_t base = inflight.front().second;
      inflight.erase(inflight.begin());
      for (int f = 0; f < r.n_frames; ++f) unpack(r, f, keypoints[base + f], descriptors[base + f]);
      return true;
    };
    for (size_t i0 = 0; i0 < n; i0 += (size_t)chunk) {
      if (inflight.size() == 3 && !drain_one()) return false;
      const int m = (int)std::min<size_t>((size_t)chunk, n - i0);
      uint8_t* stage = nullptr;
      if (gh_orb_stream_staging(batch_.s, &stage) != GH_OK) return fail("gh_orb_stream_staging");
      for (int f = 0; f < m; ++f) std::memcpy(stage + (size_t)f * fbytes, images[i0 + f].data, fbytes);
      int64_t t = -1;
      if (gh_orb_stream_submit(batch_.s, nullptr, m, &t) != GH_OK) return fail("gh_orb_stream_submit");
      inflight.push_back(std::make_pair(t, i0));
    }
    while (!inflight.empty())
      if (!drain_one()) return false;
    return true;
  }

  long submit(const GSLAM::GImage& image) override {
    std::lock_guard<std::mutex> lock(mu_);
    if (image.empty() || image.elemSize1() != 1 || !context()) return -1;
    if (!ensure_stream(async_, image, 1, kAsyncDepth)) return -1;
    int64_t t = -1;
    // the image is read by the DMA (pinned memory) or staged by the runtime (pageable) before submit returns control of
    // it only in the second case, so copy it into the slot's pinned block: the caller may free the GImage at once
    uint8_t* stage = nullptr;
    if (gh_orb_stream_staging(async_.s, &stage) != GH_OK) { fail("gh_orb_stream_staging"); return -1; }
    std::memcpy(stage, image.data, async_.frame_bytes);
    if (gh_orb_stream_submit(async_.s, nullptr, 1, &t) != GH_OK) { fail("gh_orb_stream_submit"); return -1; }
    return (long)t;
  }

  bool collect(long ticket, std::vector<GSLAM::KeyPoint>& keypoints, GSLAM::GImage& descriptors) override {
    gh_orb_stream* s;
    {
      std::lock_guard<std::mutex> lock(mu_);
      s = async_.s;
    }
    if (!s) return false;
    gh_orb_stream_result r;
    if (gh_orb_stream_collect(s, (int64_t)ticket, &r) != GH_OK) return fail("gh_orb_stream_collect");  // blocks WITHOUT mu_
    unpack(r, 0, keypoints, descriptors);
    return true;
  }

  int asyncDepth() const override { return kAsyncDepth; }

  bool match(const GSLAM::GImage& q, const GSLAM::GImage& t, std::vector<std::pair<int, int> >& matches,
             std::vector<uchar>* mask = NULL) override {
    std::lock_guard<std::mutex> lock(mu_);
    matches.clear();
    if (mask) mask->clear();
    // descriptor width as the reference's DistanceFactory::create picks the distance by it (GSLAM/core/Vocabulary.h:565-567):
    // 32 bytes (ORB: hamming32), 64 (hamming64), other multiples of 8 up to 256 (hamming8x)
    const int nb = (int)(q.cols * q.elemSize());
    if (nb < 8 || nb > 256 || nb % 8 != 0 || (t.rows > 0 && (int)(t.cols * t.elemSize()) != nb) || !context()) return false;
    const int nq = q.rows, nt = t.rows;
    if (nq == 0) return true;
    if (nb != 32 && (nt > 65535 || nq > 65535)) return false;
    std::vector<int32_t> idx1((size_t)nq), back;
    std::vector<uint16_t> d1((size_t)nq), d2((size_t)nq), bd1, bd2;
    if (gh_bf_match_bytes_host(ctx_, q.data, nq, t.data, nt, nb, idx1.data(), d1.data(), d2.data()) != GH_OK)
      return fail("gh_bf_match_bytes_host");
    if (_config.matchCrossCheck && nt > 0) {
      back.resize((size_t)nt); bd1.resize((size_t)nt); bd2.resize((size_t)nt);
      if (gh_bf_match_bytes_host(ctx_, t.data, nt, q.data, nq, nb, back.data(), bd1.data(), bd2.data()) != GH_OK)
        return fail("gh_bf_match_bytes_host(back)");
    }
    std::vector<uchar> keep((size_t)nq, 0);
    for (int i = 0; i < nq; ++i) {  // same integer rules as gh_match_mask_dev (tiny, host side here)
      const int j = idx1[i];
      bool ok = j >= 0 && (int)d1[i] <= _config.matchMaxDistance;
      if (ok && _config.matchRatioNum > 0) ok = (int)d1[i] * _config.matchRatioDen < _config.matchRatioNum * (int)d2[i];
      if (ok && _config.matchCrossCheck) ok = j < nt && back[j] == i;
      keep[i] = ok ? 1 : 0;
      if (ok) matches.push_back(std::make_pair(i, j));
    }
    if (mask) mask->swap(keep);
    return true;
  }

  // All pairs as ONE batched device call: the descriptor matrices go up once as F x cap x 32 (cap = the largest frame), the
  // forward rows -- and, with cross-checking, the backward rows of the swapped pairs -- come from gh_bf_match_pairs_dev
  // (which runs the exact MFMA kernel once the batch is large), the configured tests from gh_match_mask_dev.
  bool matchBatch(const std::vector<GSLAM::GImage>& descriptors, const std::vector<std::pair<int, int> >& pairs,
                  std::vector<std::vector<std::pair<int, int> > >& matches) override {
    std::lock_guard<std::mutex> lock(mu_);
    matches.assign(pairs.size(), std::vector<std::pair<int, int> >());
    if (pairs.empty()) return true;
    if (!context()) return false;
    const int F = (int)descriptors.size(), P = (int)pairs.size();
    int cap = 1;
    for (const GSLAM::GImage& d : descriptors) {
      if (d.rows > 0 && d.cols * d.elemSize() != 32) return false;
      cap = std::max(cap, d.rows);
    }
    cap = (cap + 15) & ~15;
    if (cap > 65535) cap = 65535;   // (a frame of 65521 .. 65535 rows: match() takes it, so must the batch)
    for (const GSLAM::GImage& d : descriptors)
      if (d.rows > 65535) return false;  // (frames, not maps: the batched entry keeps 16-bit train indices)
    const bool cross = _config.matchCrossCheck;
    const int NP = cross ? 2 * P : P;
    std::vector<int32_t> counts((size_t)F), pq((size_t)NP), pt((size_t)NP);
    for (int f = 0; f < F; ++f) counts[f] = descriptors[f].rows;
    for (int p = 0; p < P; ++p) {
      if (pairs[p].first < 0 || pairs[p].second < 0 || pairs[p].first >= F || pairs[p].second >= F) return false;
      pq[p] = pairs[p].first;
      pt[p] = pairs[p].second;
      if (cross) {
        pq[P + p] = pairs[p].second;
        pt[P + p] = pairs[p].first;
      }
    }
    const size_t desc_b = (size_t)F * cap * 32, cnt_b = ((size_t)F * 4 + 255) & ~(size_t)255, pair_b = ((size_t)NP * 4 + 255) & ~(size_t)255,
                 idx_b = (size_t)NP * cap * 4, d_b = (size_t)NP * cap * 2, keep_b = ((size_t)P * cap + 255) & ~(size_t)255;
    const size_t total = desc_b + cnt_b + 2 * pair_b + idx_b + 2 * d_b + keep_b;
    void* dev = nullptr;
    if (gh_dev_alloc(ctx_, total, &dev) != GH_OK) return fail("gh_dev_alloc");
    uint8_t* b = (uint8_t*)dev;
    uint8_t* d_desc = b;
    int32_t* d_cnt = (int32_t*)(b + desc_b);
    int32_t* d_pq = (int32_t*)(b + desc_b + cnt_b);
    int32_t* d_pt = (int32_t*)(b + desc_b + cnt_b + pair_b);
    int32_t* d_idx = (int32_t*)(b + desc_b + cnt_b + 2 * pair_b);
    uint16_t* d_d1 = (uint16_t*)((uint8_t*)d_idx + idx_b);
    uint16_t* d_d2 = (uint16_t*)((uint8_t*)d_d1 + d_b);
    uint8_t* d_keep = (uint8_t*)d_d2 + d_b;
    bool ok = true;
    {
      std::vector<uint8_t> host(desc_b, 0);
      for (int f = 0; f < F; ++f)
        if (descriptors[f].rows > 0) std::memcpy(&host[(size_t)f * cap * 32], descriptors[f].data, (size_t)descriptors[f].rows * 32);
      ok = gh_dev_upload(ctx_, d_desc, host.data(), desc_b) == GH_OK && gh_dev_upload(ctx_, d_cnt, counts.data(), (size_t)F * 4) == GH_OK &&
           gh_dev_upload(ctx_, d_pq, pq.data(), (size_t)NP * 4) == GH_OK && gh_dev_upload(ctx_, d_pt, pt.data(), (size_t)NP * 4) == GH_OK;
    }
    ok = ok && gh_bf_match_pairs_dev(ctx_, d_desc, d_cnt, cap, d_pq, d_pt, NP, d_idx, d_d1, d_d2) == GH_OK;
    for (int p = 0; p < P && ok; ++p) {
      const int nq = counts[pq[p]], nt = counts[pt[p]];
      if (nq == 0) continue;
      ok = gh_match_mask_dev(ctx_, d_idx + (size_t)p * cap, d_d1 + (size_t)p * cap, d_d2 + (size_t)p * cap, nq,
                             cross ? d_idx + (size_t)(P + p) * cap : NULL, nt, _config.matchMaxDistance, _config.matchRatioNum,
                             _config.matchRatioDen, cross ? 1 : 0, d_keep + (size_t)p * cap) == GH_OK;
    }
    std::vector<int32_t> idx((size_t)P * cap);
    std::vector<uint8_t> keep((size_t)P * cap);
    ok = ok && gh_dev_download(ctx_, idx.data(), d_idx, (size_t)P * cap * 4) == GH_OK &&
         gh_dev_download(ctx_, keep.data(), d_keep, (size_t)P * cap) == GH_OK;
    gh_dev_free(ctx_, dev);
    if (!ok) return fail("matchBatch");
    for (int p = 0; p < P; ++p) {
      const int nq = counts[pq[p]];
      for (int i = 0; i < nq; ++i)
        if (keep[(size_t)p * cap + i]) matches[p].push_back(std::make_pair(i, (int)idx[(size_t)p * cap + i]));
    }
    return true;
  }

 private:
  enum { kAsyncDepth = 8 };
  struct StreamSlot {
    gh_orb_stream* s = nullptr;
    int w = 0, h = 0, ch = 0, chunk = 0, k = 0, l = 0, ini = 0, mn = 0;
    size_t frame_bytes = 0;
  };
  StreamSlot batch_, async_;
  bool ensure_stream(StreamSlot& ss, const GSLAM::GImage& img, int chunk, int depth) {
    const int w = img.cols, h = img.rows, ch = img.channels();
    if (ch != 1 && ch != 3 && ch != 4) return false;
    if (ss.s && ss.w == w && ss.h == h && ss.ch == ch && ss.chunk >= chunk && ss.k == _config.nFeatures && ss.l == _config.nLevels &&
        ss.ini == _config.iniThFAST && ss.mn == _config.minThFAST)
      return true;
    if (ss.s) gh_orb_stream_destroy(ss.s);
    ss.s = nullptr;
    gh_orb_params p;
    gh_orb_default_params(&p);
    p.n_features = _config.nFeatures;
    p.n_levels = _config.nLevels;
    p.ini_th_fast = _config.iniThFAST;
    p.min_th_fast = _config.minThFAST;
    const size_t fbytes = (size_t)w * h * ch;
    if (gh_orb_stream_create(ctx_, w, h, ch, w * ch, fbytes, chunk, depth, &p, &ss.s) != GH_OK) {
      ss.s = nullptr;
      return fail("gh_orb_stream_create");
    }
    if (!configure(gh_orb_stream_plan(ss.s))) return false;
    ss.w = w; ss.h = h; ss.ch = ch; ss.chunk = chunk; ss.k = p.n_features; ss.l = p.n_levels; ss.ini = p.ini_th_fast; ss.mn = p.min_th_fast;
    ss.frame_bytes = fbytes;
    return true;
  }
  // svar FeatureDetectorHIP.Steering = 1: continuous orientation + per-keypoint pattern rotation (OpenCV / ORB-SLAM steering,
  // gh_orb_plan_set_steering); svar FeatureDetectorHIP.Pattern = <text file of 1024 integers>: the 256 x 4 test pattern
  // (e.g. the canonical bit_pattern_31_ copied from OpenCV's orb.cpp, which needs Steering = 1: it reaches radius 18.4)
  bool configure(gh_orb_plan* plan) {
    if (!plan) return false;
    const int steer = svar.GetInt("FeatureDetectorHIP.Steering", 0);
    if (steer != 0 && gh_orb_plan_set_steering(plan, 1) != GH_OK) return fail("gh_orb_plan_set_steering");
    // svar FeatureDetectorHIP.Distribution = 1: ORB-SLAM's per-cell FAST + quadtree (gh_orb_plan_set_distribution)
    if (svar.GetInt("FeatureDetectorHIP.Distribution", 0) != 0 && gh_orb_plan_set_distribution(plan, 1) != GH_OK)
      return fail("gh_orb_plan_set_distribution");
    const std::string pf = svar.GetString("FeatureDetectorHIP.Pattern", "");
    if (!pf.empty()) {
      std::ifstream in(pf.c_str());
      std::vector<int8_t> pat;
      int v;
      char c;
      bool in_range = true;
      while (in) {
        if (in >> v) {
          in_range = in_range && v >= -128 && v <= 127;  // (a number of a comment or an identifier pasted along with the table)
          pat.push_back((int8_t)v);
        } else if (!in.eof()) { in.clear(); in >> c; }  // separators: commas, braces, comments' punctuation
      }
      if (!in_range) {
        LOG(ERROR) << "FeatureDetectorHIP: " << pf << " holds an integer outside [-128, 127]: not a test pattern (strip comments and identifiers)";
        return false;
      }
      if (pat.size() != 1024) {
        LOG(ERROR) << "FeatureDetectorHIP: " << pf << " holds " << pat.size() << " integers, a test pattern has 1024";
        return false;
      }
      if (gh_orb_plan_set_pattern(plan, pat.data()) != GH_OK) return fail("gh_orb_plan_set_pattern");
    }
    return true;
  }
  static void unpack(const gh_orb_stream_result& r, int f, std::vector<GSLAM::KeyPoint>& kps, GSLAM::GImage& desc) {
    const int o = r.offsets[f], n = r.offsets[f + 1] - o;
    kps.resize((size_t)n);
    if (n > 0) std::memcpy((void*)kps.data(), r.kps + o, (size_t)n * sizeof(gh_keypoint));
    desc = GSLAM::GImage(n, 32, GSLAM::GImageType<uchar>::Type, const_cast<uchar*>(r.desc) + (size_t)o * 32, true);
  }
  void free_device_buffers() {
    if (ctx_) {
      if (d_bgr_) gh_dev_free(ctx_, d_bgr_);
      if (d_gray_) gh_dev_free(ctx_, d_gray_);
      if (d_out_) gh_dev_free(ctx_, d_out_);
    }
    d_bgr_ = d_gray_ = d_out_ = nullptr;
    bgr_cap_ = gray_cap_ = out_cap_ = 0;
  }
  bool fail(const char* what) {
    LOG(ERROR) << "FeatureDetectorHIP: " << what << " failed: " << (ctx_ ? gh_last_error(ctx_) : "no context");
    return false;
  }
  bool context() {
    if (!ctx_) {
      const int dev = svar.GetInt("FeatureDetectorHIP.Device", 0);
      if (gh_ctx_create(dev, &ctx_) != GH_OK) {
        ctx_ = nullptr;
        LOG(ERROR) << "FeatureDetectorHIP: no usable HIP device " << dev << " (there is no CPU fallback)";
      }
    }
    return ctx_ != nullptr;
  }
  gh_ctx* ctx_;
  gh_orb_plan* plan_;
  int pw_, ph_, pk_, pl_, pi_, pm_;  // geometry + parameters the cached plan was built for
  void *d_bgr_ = nullptr, *d_gray_ = nullptr, *d_out_ = nullptr;  // colour path: BGR, luma, result block
  size_t bgr_cap_ = 0, gray_cap_ = 0, out_cap_ = 0;
  std::vector<uint8_t> host_out_;
  std::mutex mu_;
};

}  // namespace

GSLAM_REGISTER_FEATUREDETECTOR(FeatureDetectorHIP);
