// libgslam_vocabulary.so — GSLAM::Vocabulary whose image transforms run on the MI355X.
//
// The reference has no Vocabulary plugin ABI: a replacement is a subclass overriding the virtual
// transform(...) methods (GSLAM/core/Vocabulary.h:174-200).  VocabularyHIP keeps every other member of the base
// class (load/save, node tables, scoring object) and only moves
//     transform(const TinyMat& features, BowVector&, FeatureVector&, int levelsup)   (:1558-1621)
//     transform(const TinyMat& features, BowVector&)                                (:1437-1497)
// to gh_bow_transform_host.  Factory (same idiom as createOptimizerInstance):
//     extern "C" std::shared_ptr<GSLAM::Vocabulary> createVocabularyInstance(const char* gbow_file)
// Results are bit-identical to the base class (tests/test_plugins_gpu.py); no GPU => falls back to nothing: the
// methods leave the outputs empty and log an error.
// Vocabulary::score is a non-virtual inline that this snapshot declares (:203-211) but never defines -- callers use
// m_scoring_object->score -- and scoring ONE pair on a GPU would be pure latency, so the
// batched form a loop detector needs is offered beside it:
//     extern "C" bool scoreVocabularyBatch(const GSLAM::Vocabulary*, const BowVector* query,
//                                          const BowVector* const* db, int n_db, double* scores)
// = m_scoring_object->score(*query, *db[j]) for every j (GSLAM/core/Vocabulary.h:691-979) through gh_bow_score_host.
#include <GSLAM/core/GSLAM.h>
#include <GSLAM/core/Vocabulary.h>

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
    if (m_nodes.empty() || m_nodeDescriptors.cols * m_nodeDescriptors.elemSize() != 32) {
      LOG(ERROR) << "VocabularyHIP: only 32-byte binary descriptors are supported";
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
    if (gh_bow_vocab_create(ctx_, m_k, m_L, (int)m_weighting, (int)m_scoring, (uint32_t)m_nodes.size(), m_nodes.data(),
                            m_nodeDescriptors.data, &voc_) != GH_OK) {
      LOG(ERROR) << "VocabularyHIP: " << gh_last_error(ctx_);
      return false;
    }
    return true;
  }

  void run(const GSLAM::TinyMat& features, int levelsup, GSLAM::BowVector* v, GSLAM::FeatureVector* fv) const {
    std::lock_guard<std::mutex> lock(mu_);
    const int n = features.rows;
    if (!voc_ || n <= 0 || features.cols * features.elemSize() != 32) return;
    if (n > 16384) {  // one workgroup sorts an image's word ids in LDS: 16384 features is the library's limit
      LOG(ERROR) << "VocabularyHIP: " << n << " features in one image exceed the limit of 16384; outputs left empty";
      return;
    }
    std::vector<uint32_t> word(n), node(n), bw(n);
    std::vector<float> weight(n), bv(n);
    int32_t nb = 0;
    if (gh_bow_transform_host(voc_, features.data, n, levelsup, word.data(), weight.data(), node.data(), bw.data(),
                              bv.data(), &nb) != GH_OK) {
      LOG(ERROR) << "VocabularyHIP: " << gh_last_error(ctx_);
      return;
    }
    GSLAM::BowVector::iterator hint = v->end();
    for (int i = 0; i < nb; ++i) hint = v->insert(hint, GSLAM::BowVector::value_type(bw[i], bv[i]));  // ascending ids
    if (fv)
      for (int i = 0; i < n; ++i)
        if (weight[i] > 0) (*fv)[node[i]].push_back((unsigned int)i);  // feature order == the reference's order
  }

  gh_ctx* ctx_;
  gh_bow_vocab* voc_;
  mutable std::mutex mu_;
};

}  // namespace

extern "C" bool scoreVocabularyBatch(const GSLAM::Vocabulary* voc, const GSLAM::BowVector* query,
                                     const GSLAM::BowVector* const* db, int n_db, double* scores) {
  const VocabularyHIP* v = dynamic_cast<const VocabularyHIP*>(voc);
  return v && query && v->scoreBatch(*query, db, n_db, scores);
}

extern "C" std::shared_ptr<GSLAM::Vocabulary> createVocabularyInstance(const char* gbow_file) {
  std::shared_ptr<VocabularyHIP> v(new VocabularyHIP());
  if (!gbow_file || !v->loadFile(gbow_file)) return std::shared_ptr<GSLAM::Vocabulary>();
  return v;
}
