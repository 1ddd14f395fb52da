// libgslam_vocabulary.so — GSLAM::Vocabulary whose image transforms run on the MI355X.
//
// The reference has no Vocabulary plugin ABI: a replacement is a subclass overriding the virtual
// transform(...) methods (GSLAM/core/Vocabulary.h:174-200).  VocabularyHIP keeps every other member of the base
// class (load/save, node tables, scoring object) and only moves
//     transform(const TinyMat& features, BowVector&, FeatureVector&, int levelsup)   (:1558-1621)
//     transform(const TinyMat& features, BowVector&)                                (:1437-1497)
// to gh_bow_transform_host.  Binary vocabularies of any width the reference's DistanceFactory accepts (:560-568: 32 bytes ->
// hamming32, 64 -> hamming64, other multiples of 8 -> hamming8x) are taken, and float vocabularies whose dimension is a
// multiple of 8 (:569-578: l2generic whatever ISA the host was built for); other float dimensions are refused at load.
// Factory (same idiom as createOptimizerInstance):
//     extern "C" std::shared_ptr<GSLAM::Vocabulary> createVocabularyInstance(const char* gbow_file)
// as well as the std::vector<TinyMat> overload (:183, one feature per element) and the single-feature
// transform(const TinyMat&) -> WordId (:200).  Images of more than 16384 features are descended on the GPU in chunks and
// merged on the host exactly as the reference accumulates them (:1572-1618).
// Results are bit-identical to the base class (tests/test_plugins_gpu.py).  There is no CPU fallback: when a transform
// cannot run (no device, library error) the outputs stay empty, an error is logged AND the failure is counted --
// the reference's transform cannot fail, so a caller that wants to notice asks
//     extern "C" int vocabularyFailureCount(const GSLAM::Vocabulary*)
// Vocabulary::score is a non-virtual inline that this snapshot declares (:203-211) but never defines -- callers use
// m_scoring_object->score -- and scoring ONE pair on a GPU would be pure latency, so the
// batched form a loop detector needs is offered beside it:
//     extern "C" bool scoreVocabularyBatch(const GSLAM::Vocabulary*, const BowVector* query,
//                                          const BowVector* const* db, int n_db, double* scores)
// = m_scoring_object->score(*query, *db[j]) for every j (GSLAM/core/Vocabulary.h:691-979) through gh_bow_score_host.
#include <GSLAM/core/GSLAM.h>
#include <GSLAM/core/Vocabulary.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

#include "gslam_hip.h"

namespace {

class VocabularyHIP : public GSLAM::Vocabulary {
 public:
  VocabularyHIP() : ctx_(nullptr), voc_(nullptr) {}
  ~VocabularyHIP() {
    if (voc_) gh_bow_vocab_destroy(voc_);
    if (ctx_) gh_ctx_destroy(ctx_);
  }

  bool loadFile(const std::string& path) { return GSLAM::Vocabulary::load(path) && upload(); }

  void transform(const GSLAM::TinyMat& features, GSLAM::BowVector& v, GSLAM::FeatureVector& fv,
                 int levelsup = 0) const override {
    v.clear();
    fv.clear();
    run(features, levelsup, &v, &fv);
  }

  void transform(const GSLAM::TinyMat& features, GSLAM::BowVector& v) const override {
    v.clear();
    run(features, 0, &v, nullptr);
  }

  // one feature per element (GSLAM/core/Vocabulary.h:183,1625-1690: the same accumulation as the matrix form)
  void transform(const std::vector<GSLAM::TinyMat>& features, GSLAM::BowVector& v, GSLAM::FeatureVector& fv,
                 int levelsup = 0) const override {
    v.clear();
    fv.clear();
    if (features.empty()) return;
    std::vector<uchar> packed(features.size() * (size_t)width_);
    for (size_t i = 0; i < features.size(); ++i) {
      if (features[i].rows < 1 || (int)(features[i].cols * features[i].elemSize()) != width_) {
        failed("a feature of the list is not one descriptor row of the vocabulary's width");
        return;
      }
      std::memcpy(&packed[i * width_], features[i].data, width_);
    }
    GSLAM::TinyMat all((int)features.size(), float_ ? width_ / 4 : width_,
                       float_ ? (int)GSLAM::GImageType<float>::Type : (int)GSLAM::GImageType<uchar>::Type, packed.data(), false);
    run(all, levelsup, &v, &fv);
  }

  // a single feature -> its word (GSLAM/core/Vocabulary.h:200,1421-1433)
  GSLAM::WordId transform(const GSLAM::TinyMat& feature) const override {
    if (empty() || feature.rows < 1 || (int)(feature.cols * feature.elemSize()) != width_) return 0;
    std::lock_guard<std::mutex> lock(mu_);
    if (!voc_) return 0;
    uint32_t word = 0, node = 0, bw = 0;
    float weight = 0, bv = 0;
    int32_t nb = 0;
    if (gh_bow_transform_host(voc_, feature.data, 1, 0, &word, &weight, &node, &bw, &bv, &nb) != GH_OK) {
      failed(gh_last_error(ctx_));
      return 0;
    }
    return (GSLAM::WordId)word;
  }

  int failures() const { return failures_; }

 public:
  // scores[j] = m_scoring_object->score(query, *db[j]); ids above 2^32 - 2 cannot occur (node ids are 32-bit in .gbow)
  bool scoreBatch(const GSLAM::BowVector& query, const GSLAM::BowVector* const* db, int n_db, double* scores) const {
    std::lock_guard<std::mutex> lock(mu_);
    if (!ctx_ || n_db < 0 || (n_db > 0 && (!db || !scores))) return false;
    std::vector<uint32_t> qi, di;
    std::vector<float> qv, dv;
    std::vector<int64_t> off((size_t)n_db + 1, 0);
    qi.reserve(query.size());
    qv.reserve(query.size());
    for (GSLAM::BowVector::const_iterator it = query.begin(); it != query.end(); ++it) {
      qi.push_back((uint32_t)it->first);
      qv.push_back(it->second);
    }
    for (int j = 0; j < n_db; ++j) {
      if (!db[j]) return false;
      for (GSLAM::BowVector::const_iterator it = db[j]->begin(); it != db[j]->end(); ++it) {
        di.push_back((uint32_t)it->first);
        dv.push_back(it->second);
      }
      off[(size_t)j + 1] = (int64_t)di.size();
    }
    if (gh_bow_score_host(ctx_, (int)m_scoring, qi.data(), qv.data(), (int)qi.size(), di.data(), dv.data(), off.data(), n_db,
                          scores) != GH_OK) {
      LOG(ERROR) << "VocabularyHIP: " << gh_last_error(ctx_);
      return false;
    }
    return true;
  }

 private:
  bool upload() {
    std::lock_guard<std::mutex> lock(mu_);
    width_ = (int)(m_nodeDescriptors.cols * m_nodeDescriptors.elemSize());
    float_ = m_nodeDescriptors.type() == GSLAM::GImageType<float>::Type;
    const bool binary = m_nodeDescriptors.type() == GSLAM::GImageType<uchar>::Type;
    if (m_nodes.empty() || (!binary && !float_) || width_ < 8 || (binary && width_ % 8 != 0) || (float_ && m_nodeDescriptors.cols % 8 != 0)) {
      LOG(ERROR) << "VocabularyHIP: binary descriptors of a multiple of 8 bytes or float descriptors of a multiple of 8 dimensions only";
      return false;
    }
    if (!ctx_ && gh_ctx_create(svar.GetInt("VocabularyHIP.Device", 0), &ctx_) != GH_OK) {
      ctx_ = nullptr;
      LOG(ERROR) << "VocabularyHIP: no usable HIP device (there is no CPU fallback)";
      return false;
    }
    if (voc_) gh_bow_vocab_destroy(voc_);
    voc_ = nullptr;
    static_assert(sizeof(Node) == 8, "Vocabulary::Node layout");
    const gh_status st =
        float_ ? gh_bow_vocab_create_f32(ctx_, m_k, m_L, (int)m_weighting, (int)m_scoring, (uint32_t)m_nodes.size(), m_nodes.data(),
                                         (const float*)m_nodeDescriptors.data, m_nodeDescriptors.cols, &voc_)
               : gh_bow_vocab_create_bytes(ctx_, m_k, m_L, (int)m_weighting, (int)m_scoring, (uint32_t)m_nodes.size(), m_nodes.data(),
                                           m_nodeDescriptors.data, width_, &voc_);
    if (st != GH_OK) {
      LOG(ERROR) << "VocabularyHIP: " << gh_last_error(ctx_);
      return false;
    }
    return true;
  }

  void failed(const char* why) const {
    ++failures_;
    LOG(ERROR) << "VocabularyHIP: transform failed, outputs left empty: " << why;
  }

  void run(const GSLAM::TinyMat& features, int levelsup, GSLAM::BowVector* v, GSLAM::FeatureVector* fv) const {
    std::lock_guard<std::mutex> lock(mu_);
    const int n = features.rows;
    if (n <= 0) return;
    if (!voc_) return failed("no device vocabulary (load failed or no GPU)");
    if ((int)(features.cols * features.elemSize()) != width_) return failed("descriptors do not have the vocabulary's width");
    constexpr int kMax = 16384;  // one workgroup sorts an image's word ids in LDS: the library's limit per call
    std::vector<uint32_t> word(n), node(n), bw(std::min(n, kMax));
    std::vector<float> weight(n), bv(std::min(n, kMax));
    int32_t nb = 0;
    for (int c0 = 0; c0 < n; c0 += kMax) {
      const int m = std::min(kMax, n - c0);
      if (gh_bow_transform_host(voc_, features.data + (size_t)c0 * width_, m, levelsup, word.data() + c0, weight.data() + c0,
                                node.data() + c0, bw.data(), bv.data(), &nb) != GH_OK)
        return failed(gh_last_error(ctx_));
    }
    if (n <= kMax) {
      GSLAM::BowVector::iterator hint = v->end();
      for (int i = 0; i < nb; ++i) hint = v->insert(hint, GSLAM::BowVector::value_type(bw[i], bv[i]));  // ascending ids
    } else {
      // more than one chunk: the per-feature words / weights came from the GPU, the image-level accumulation is the
      // reference's own sequence (Vocabulary.h:1572-1618): addWeight / addIfNotExist in feature order, / size, normalize
      LNorm norm;  // (members of the base class)
      const bool must = m_scoring_object->mustNormalize(norm);
      const bool tf = m_weighting == TF || m_weighting == TF_IDF;
      for (int i = 0; i < n; ++i) {
        if (!(weight[i] > 0)) continue;
        if (tf) addWeight(*v, word[i], weight[i]);
        else addIfNotExist(*v, word[i], weight[i]);
      }
      if (tf && !v->empty() && !must) {
        const double nd = v->size();
        for (GSLAM::BowVector::iterator it = v->begin(); it != v->end(); ++it) it->second /= nd;
      }
      if (must) normalize(*v, norm);
    }
    if (fv)
      for (int i = 0; i < n; ++i)
        if (weight[i] > 0) (*fv)[node[i]].push_back((unsigned int)i);  // feature order == the reference's order
  }

  gh_ctx* ctx_;
  gh_bow_vocab* voc_;
  int width_ = 32;  // descriptor bytes
  bool float_ = false;
  mutable std::mutex mu_;
  mutable std::atomic<int> failures_{0};
};

}  // namespace

extern "C" bool scoreVocabularyBatch(const GSLAM::Vocabulary* voc, const GSLAM::BowVector* query,
                                     const GSLAM::BowVector* const* db, int n_db, double* scores) {
  const VocabularyHIP* v = dynamic_cast<const VocabularyHIP*>(voc);
  return v && query && v->scoreBatch(*query, db, n_db, scores);
}

extern "C" int vocabularyFailureCount(const GSLAM::Vocabulary* voc) {
  const VocabularyHIP* v = dynamic_cast<const VocabularyHIP*>(voc);
  return v ? v->failures() : -1;
}

extern "C" std::shared_ptr<GSLAM::Vocabulary> createVocabularyInstance(const char* gbow_file) {
  std::shared_ptr<VocabularyHIP> v(new VocabularyHIP());
  if (!gbow_file || !v->loadFile(gbow_file)) return std::shared_ptr<GSLAM::Vocabulary>();
  return v;
}
