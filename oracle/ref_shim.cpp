// ref_shim.cpp — thin C wrapper around the REFERENCE's own code, compiled from the sources where
// they lie under /root/reference (never copied).  Output: oracle/_ref/libgslam_ref.so.
// Used (a) to pin the oracle (tests + tools/gen_golden.py) and (b) as the "reference" CPU baseline
// for BF matching.  Built with the reference's flags (-O3 -DNDEBUG, CMakeLists.txt:9-11); a second
// build adds -mpopcnt (libgslam_ref_popcnt.so).
#include <GSLAM/core/GSLAM.h>
#include <GSLAM/core/Vocabulary.h>

#include <cfloat>
#include <cstdint>
#include <cstring>

extern "C" {

float ref_hamming32(const unsigned char* a, const unsigned char* b) {
  return GSLAM::Vocabulary::DistanceFactory::hamming32(a, b);
}

// Brute-force loop written exactly like Vocabulary::transform's child scan
// (GSLAM/core/Vocabulary.h:1712-1725): FLT_MAX start, strict '<'.
void ref_bf_match(const unsigned char* q, int nq, const unsigned char* t, int nt, int32_t* idx1, float* d1,
                  int threads) {
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int i = 0; i < nq; ++i) {
    float best_d = std::numeric_limits<float>::max();
    int best = -1;
    for (int j = 0; j < nt; ++j) {
      float d = GSLAM::Vocabulary::DistanceFactory::hamming32(q + (size_t)i * 32, t + (size_t)j * 32);
      if (d < best_d) {
        best_d = d;
        best = j;
      }
    }
    idx1[i] = best;
    d1[i] = best_d;
  }
}

// the reference's own wider Hamming distances (GSLAM/core/Vocabulary.h:493-513)
float ref_hamming64(const unsigned char* a, const unsigned char* b) { return GSLAM::Vocabulary::DistanceFactory::hamming64(a, b); }
float ref_hamming8x(const unsigned char* a, const unsigned char* b, int bytes) {
  return GSLAM::Vocabulary::DistanceFactory::hamming8x(a, b, bytes);
}
// ... and the child-scan loop over rows of `bytes` bytes: hamming64 for 64, hamming8x otherwise (what
// Vocabulary picks for a descriptor width, DistanceFactory::create, Vocabulary.h:565-567)
void ref_bf_match_bytes(const unsigned char* q, int nq, const unsigned char* t, int nt, int bytes, int32_t* idx1, float* d1) {
  for (int i = 0; i < nq; ++i) {
    float best_d = std::numeric_limits<float>::max();
    int best = -1;
    for (int j = 0; j < nt; ++j) {
      const unsigned char *a = q + (size_t)i * bytes, *b = t + (size_t)j * bytes;
      const float d = bytes == 64 ? GSLAM::Vocabulary::DistanceFactory::hamming64(a, b)
                                  : GSLAM::Vocabulary::DistanceFactory::hamming8x(a, b, bytes);
      if (d < best_d) {
        best_d = d;
        best = j;
      }
    }
    idx1[i] = best;
    d1[i] = best_d;
  }
}

// Lie-group helpers of the BA pose update (GSLAM/core/SE3.h, SO3.h).  Pose layout: tx ty tz qx qy qz qw.
void ref_se3_exp(const double* xi6, double* pose7) {
  GSLAM::Vector<double, 6> l;
  for (int i = 0; i < 6; ++i) l[i] = xi6[i];
  GSLAM::SE3 T = GSLAM::SE3::exp(l);
  auto t = T.get_translation();
  auto r = T.get_rotation();
  pose7[0] = t.x; pose7[1] = t.y; pose7[2] = t.z;
  pose7[3] = r.x; pose7[4] = r.y; pose7[5] = r.z; pose7[6] = r.w;
}

void ref_se3_log(const double* pose7, double* xi6) {
  GSLAM::SE3 T(pose7[0], pose7[1], pose7[2], pose7[3], pose7[4], pose7[5], pose7[6]);
  auto l = T.log();
  for (int i = 0; i < 6; ++i) xi6[i] = l[i];
}

void ref_se3_mul(const double* a7, const double* b7, double* out7) {
  GSLAM::SE3 A(a7[0], a7[1], a7[2], a7[3], a7[4], a7[5], a7[6]);
  GSLAM::SE3 B(b7[0], b7[1], b7[2], b7[3], b7[4], b7[5], b7[6]);
  GSLAM::SE3 C = A * B;
  auto t = C.get_translation();
  auto r = C.get_rotation();
  out7[0] = t.x; out7[1] = t.y; out7[2] = t.z;
  out7[3] = r.x; out7[4] = r.y; out7[5] = r.z; out7[6] = r.w;
}

void ref_se3_inverse(const double* a7, double* out7) {
  GSLAM::SE3 A(a7[0], a7[1], a7[2], a7[3], a7[4], a7[5], a7[6]);
  GSLAM::SE3 C = A.inverse();
  auto t = C.get_translation();
  auto r = C.get_rotation();
  out7[0] = t.x; out7[1] = t.y; out7[2] = t.z;
  out7[3] = r.x; out7[4] = r.y; out7[5] = r.z; out7[6] = r.w;
}

void ref_se3_apply(const double* a7, const double* p3, double* out3) {
  GSLAM::SE3 A(a7[0], a7[1], a7[2], a7[3], a7[4], a7[5], a7[6]);
  GSLAM::Point3d p(p3[0], p3[1], p3[2]);
  GSLAM::Point3d o = A * p;
  out3[0] = o.x; out3[1] = o.y; out3[2] = o.z;
}

// SIM3 algebra of the pose-graph / alignment oracle (GSLAM/core/SIM3.h:114-270).  Layout: qx qy qz qw tx ty tz s.
static GSLAM::SIM3 sim3_from(const double* s8) {
  return GSLAM::SIM3(GSLAM::SO3(s8[0], s8[1], s8[2], s8[3]), GSLAM::Point3d(s8[4], s8[5], s8[6]), s8[7]);
}
static void sim3_to(const GSLAM::SIM3& S, double* o8) {
  auto r = S.get_rotation();
  auto t = S.get_translation();
  o8[0] = r.x; o8[1] = r.y; o8[2] = r.z; o8[3] = r.w;
  o8[4] = t.x; o8[5] = t.y; o8[6] = t.z; o8[7] = S.get_scale();
}
void ref_sim3_exp(const double* mu7, double* out8) {
  GSLAM::Vector<double, 7> m;
  for (int i = 0; i < 7; ++i) m[i] = mu7[i];
  sim3_to(GSLAM::SIM3::exp(m), out8);
}
void ref_sim3_log(const double* s8, double* mu7) {
  auto l = sim3_from(s8).log();
  for (int i = 0; i < 7; ++i) mu7[i] = l[i];
}
void ref_sim3_mul(const double* a8, const double* b8, double* out8) { sim3_to(sim3_from(a8) * sim3_from(b8), out8); }
void ref_sim3_apply(const double* a8, const double* p3, double* out3) {
  GSLAM::Point3d o = sim3_from(a8) * GSLAM::Point3d(p3[0], p3[1], p3[2]);
  out3[0] = o.x; out3[1] = o.y; out3[2] = o.z;
}

int ref_sizeof_keypoint() { return (int)sizeof(GSLAM::KeyPoint); }
int ref_sizeof_se3() { return (int)sizeof(GSLAM::SE3); }
int ref_sizeof_sim3() { return (int)sizeof(GSLAM::SIM3); }
}

// ---------------------------------------------------------------- Vocabulary
// The reference's own BoW transform: GSLAM::Vocabulary::load(std::istream&) (Vocabulary.h:1891-1932) on an
// in-memory .gbow image, then transform(features, bow, fv, levelsup) (:1558-1621).  Flat outputs:
//   bow_ids/bow_vals (ascending word id, map order), fv pairs (node id, feature index) in map order.
#include <sstream>

extern "C" {

void* ref_vocab_load(const unsigned char* gbow, size_t bytes) {
  std::string buf((const char*)gbow, bytes);
  std::istringstream is(buf, std::ios::binary);
  GSLAM::Vocabulary* v = new GSLAM::Vocabulary();
  if (!v->load(is)) {
    delete v;
    return nullptr;
  }
  return v;
}

void ref_vocab_free(void* v) { delete (GSLAM::Vocabulary*)v; }

int ref_vocab_info(void* vp, int* k, int* L, int* nnodes) {
  GSLAM::Vocabulary* v = (GSLAM::Vocabulary*)vp;
  *k = v->m_k;
  *L = v->m_L;
  *nnodes = (int)v->m_nodes.size();
  return 0;
}

// returns number of BoW entries; fv_n receives the number of (node, feature) pairs
int ref_vocab_transform_bytes(void* vp, const unsigned char* desc, int n, int levelsup, uint64_t* bow_ids, float* bow_vals,
                              uint64_t* fv_nodes, uint32_t* fv_feat, int* fv_n, int desc_bytes);
int ref_vocab_transform(void* vp, const unsigned char* desc, int n, int levelsup, uint64_t* bow_ids, float* bow_vals,
                        uint64_t* fv_nodes, uint32_t* fv_feat, int* fv_n) {
  return ref_vocab_transform_bytes(vp, desc, n, levelsup, bow_ids, bow_vals, fv_nodes, fv_feat, fv_n, 32);
}
// float descriptors of `dims` dimensions through the reference (l2generic for dims % 8 == 0)
int ref_vocab_transform_f32(void* vp, const float* desc, int n, int dims, int levelsup, uint64_t* bow_ids, float* bow_vals,
                            uint64_t* word, float* weight, uint64_t* node) {
  GSLAM::Vocabulary* v = (GSLAM::Vocabulary*)vp;
  GSLAM::TinyMat features(n, dims, GSLAM::GImageType<float>::Type, (uchar*)desc, false);
  GSLAM::BowVector bow;
  GSLAM::FeatureVector fv;
  v->transform(features, bow, fv, levelsup);
  int i = 0;
  for (auto& kv : bow) {
    bow_ids[i] = kv.first;
    bow_vals[i] = kv.second;
    ++i;
  }
  for (int f = 0; f < n; ++f) {
    GSLAM::TinyMat one(1, dims, GSLAM::GImageType<float>::Type, (uchar*)(desc + (size_t)f * dims), false);
    GSLAM::WordId id;
    GSLAM::WordValue w;
    GSLAM::NodeId nid;
    v->transform(one, id, w, &nid, levelsup);
    word[f] = id;
    weight[f] = w;
    node[f] = nid;
  }
  return i;
}
// descriptors of desc_bytes bytes (the reference picks hamming32 / hamming64 / hamming8x by the column count)
int ref_vocab_transform_bytes(void* vp, const unsigned char* desc, int n, int levelsup, uint64_t* bow_ids, float* bow_vals,
                              uint64_t* fv_nodes, uint32_t* fv_feat, int* fv_n, int desc_bytes) {
  GSLAM::Vocabulary* v = (GSLAM::Vocabulary*)vp;
  GSLAM::TinyMat features(n, desc_bytes, GSLAM::GImageType<uchar>::Type, (uchar*)desc, false);
  GSLAM::BowVector bow;
  GSLAM::FeatureVector fv;
  v->transform(features, bow, fv, levelsup);
  int i = 0;
  for (auto& kv : bow) {
    bow_ids[i] = kv.first;
    bow_vals[i] = kv.second;
    ++i;
  }
  int j = 0;
  for (auto& kv : fv)
    for (unsigned f : kv.second) {
      fv_nodes[j] = kv.first;
      fv_feat[j] = f;
      ++j;
    }
  *fv_n = j;
  return i;
}

// per-feature word / weight / node (single-feature transform, Vocabulary.h:1695-1736)
void ref_vocab_words_bytes(void* vp, const unsigned char* desc, int n, int levelsup, uint64_t* word, float* weight,
                           uint64_t* node, int desc_bytes);
void ref_vocab_words(void* vp, const unsigned char* desc, int n, int levelsup, uint64_t* word, float* weight,
                     uint64_t* node) {
  ref_vocab_words_bytes(vp, desc, n, levelsup, word, weight, node, 32);
}
void ref_vocab_words_bytes(void* vp, const unsigned char* desc, int n, int levelsup, uint64_t* word, float* weight,
                           uint64_t* node, int desc_bytes) {
  GSLAM::Vocabulary* v = (GSLAM::Vocabulary*)vp;
  for (int i = 0; i < n; ++i) {
    GSLAM::TinyMat f(1, desc_bytes, GSLAM::GImageType<uchar>::Type, (uchar*)desc + (size_t)i * desc_bytes, false);
    GSLAM::WordId id;
    GSLAM::WordValue w;
    GSLAM::NodeId nid;
    v->transform(f, id, w, &nid, levelsup);
    word[i] = id;
    weight[i] = w;
    node[i] = nid;
  }
}

double ref_vocab_score(void* vp, const uint64_t* a_ids, const float* a_vals, int na, const uint64_t* b_ids,
                       const float* b_vals, int nb) {
  GSLAM::Vocabulary* v = (GSLAM::Vocabulary*)vp;
  GSLAM::BowVector a, b;
  for (int i = 0; i < na; ++i) a[a_ids[i]] = a_vals[i];
  for (int i = 0; i < nb; ++i) b[b_ids[i]] = b_vals[i];
  return v->m_scoring_object->score(a, b);
}
}

// ---------------------------------------------------------------- Undistorter
// The reference's own remap tables (UndistorterImpl::prepareReMap, GSLAM/core/Undistorter.h:120-203) and
// its undistort / undistortFast loops (:206-348) for pinning oracle/undist_oracle.c and the golden vectors.
#include <GSLAM/core/Undistorter.h>

extern "C" {

void* ref_undist_create(const double* pin, int n_in, const double* pout, int n_out) {
  std::streambuf* old = std::cout.rdbuf(nullptr);  // prepareReMap prints the camera info
  GSLAM::UndistorterImpl* u = new GSLAM::UndistorterImpl(GSLAM::Camera(std::vector<double>(pin, pin + n_in)),
                                                         GSLAM::Camera(std::vector<double>(pout, pout + n_out)));
  std::cout.rdbuf(old);
  if (!u->valid) {
    delete u;
    return nullptr;
  }
  return u;
}
void ref_undist_free(void* u) { delete (GSLAM::UndistorterImpl*)u; }
void ref_undist_dims(void* up, int* wi, int* hi, int* wo, int* ho) {
  GSLAM::UndistorterImpl* u = (GSLAM::UndistorterImpl*)up;
  *wi = u->camera_in.width(); *hi = u->camera_in.height();
  *wo = u->camera_out.width(); *ho = u->camera_out.height();
}
void ref_undist_tables(void* up, float* remapX, float* remapY, int* remapFast, int* remapIdx, float* remapCoef) {
  GSLAM::UndistorterImpl* u = (GSLAM::UndistorterImpl*)up;
  size_t n = (size_t)u->camera_out.width() * u->camera_out.height();
  memcpy(remapX, u->remapX, n * 4);
  memcpy(remapY, u->remapY, n * 4);
  memcpy(remapFast, u->remapFast, n * 4);
  memcpy(remapIdx, u->remapIdx, n * 16);
  memcpy(remapCoef, u->remapCoef, n * 16);
}
// out must be pre-filled by the caller (the reference leaves unmapped pixels untouched in several branches)
int ref_undist_run(void* up, const unsigned char* img, int channels, int fast, unsigned char* out) {
  GSLAM::UndistorterImpl* u = (GSLAM::UndistorterImpl*)up;
  const int type = channels == 1 ? GSLAM::GImageType<uchar, 1>::Type
                                 : (channels == 3 ? GSLAM::GImageType<uchar, 3>::Type : GSLAM::GImageType<uchar, 4>::Type);
  GSLAM::GImage in(u->camera_in.height(), u->camera_in.width(), type, (uchar*)img, false);
  GSLAM::GImage res;
  bool ok = fast ? u->undistortFast(in, res) : u->undistort(in, res);
  if (!ok) return 0;
  memcpy(out, res.data, (size_t)res.total() * res.elemSize());
  return 1;
}
}

// ---------------------------------------------------------------- Camera::Project (self-calibration, Optimizer.h:169-171)
// The reference's camera models evaluated on camera-frame points (GSLAM/core/Camera.h:213-227 pinhole, :386-407 OpenCV):
// pins oracle_cam_project (graph_oracle.c) and the golden vectors tests/golden/camera_reference.npz.
extern "C" {
int ref_camera_project(const double* params, int n_params, const double* xyz, int n, double* uv) {
  GSLAM::Camera cam(std::vector<double>(params, params + n_params));
  if (!cam.isValid()) return 0;
  for (int i = 0; i < n; ++i) {
    const GSLAM::Point2d p = cam.Project(GSLAM::Point3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
    uv[2 * i] = p.x;
    uv[2 * i + 1] = p.y;
  }
  return 1;
}
}
