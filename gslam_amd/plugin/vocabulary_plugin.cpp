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

extern "C" std::shared_ptr<GSLAM::Vocabulary> createVocabularyInstance(const char* gbow_file) {
  std::shared_ptr<VocabularyHIP> v(new VocabularyHIP());
  if (!gbow_file || !v->loadFile(gbow_file)) return std::shared_ptr<GSLAM::Vocabulary>();
  return v;
}
