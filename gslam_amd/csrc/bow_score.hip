// Batched BoW scoring on gfx950: GSLAM::Vocabulary::score(a, b) = m_scoring_object->score(a, b) for every pair of a
// block of query BowVectors and a database of BowVectors (loop-closure candidate scoring: one query against every
// keyframe), SURVEY.md 8 f1.
//
// Reference semantics followed (GSLAM/core/Vocabulary.h):
//   :691-736  L1Scoring            sum over common words of |vi - wi| - |vi| - |wi|;  -s / 2
//   :741-790  L2Scoring            sum vi wi;  s >= 1 ? 1 : 1 - sqrt(1 - s)
//   :795-838  ChiSquareScoring     sum vi wi / (vi + wi) where vi + wi != 0;  2 s
//   :843-891  KLScoring            every word of v1: matched vi log(vi / wi), unmatched vi (log vi - log DBL_EPSILON)
//   :896-934  BhattacharyyaScoring sum sqrt(vi wi)
//   :939-979  DotProductScoring    sum vi wi
// WordValue is float and the reference's unqualified fabs / sqrt / log resolve to the float overloads, so each term is
// formed in single precision and accumulated into ONE double in ascending word-id order.  That order is kept here
// exactly: a wave owns one (query, database entry) pair, its lanes intersect 64 words at a time by binary search, and
// the matched terms are then added by a wave-uniform loop over the ballot mask (v_readlane), lowest lane first.  Five of
// the six scores are therefore bit-identical to the reference; KL goes through logf, where ocml and glibc may differ in
// the last ulp (tests: 1e-6 relative).
//
// Layout: the padded arrays gh_bow_transform_dev writes -- ids ascending, n valid entries per vector, capacity `cap`.
#include "common.h"

namespace {

constexpr double kLogEps = -36.043653389117154;  // log(DBL_EPSILON), GeneralScoring::LOG_EPS (:631-634)

__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// first index in [0, n) whose id is >= key (ids ascending)
template <typename P>
__device__ __forceinline__ int lower_bound_u32(P ids, int n, uint32_t key) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (ids[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

template <int SC>
__global__ __launch_bounds__(256) void bow_score_kernel(const uint32_t* __restrict__ q_word,
                                                        const float* __restrict__ q_val,
                                                        const int32_t* __restrict__ q_n, int cap_q,
                                                        const uint32_t* __restrict__ db_word,
                                                        const float* __restrict__ db_val,
                                                        const int32_t* __restrict__ db_n, int cap_db, int n_db,
                                                        double* __restrict__ out, int stage) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* s_id = reinterpret_cast<uint32_t*>(smem);
  float* s_val = reinterpret_cast<float*>(smem + (size_t)cap_q * 4);
  const int q = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + wave;
  const int nq = min(q_n[q], cap_q);
  const uint32_t* qi = q_word + (size_t)q * cap_q;
  const float* qv = q_val + (size_t)q * cap_q;
  if (stage) {  // the query vector is read by every wave of the block: stage it once
    for (int i = threadIdx.x; i < nq; i += 256) {
      s_id[i] = qi[i];
      s_val[i] = qv[i];
    }
    __syncthreads();
  }
  if (j >= n_db) return;
  const int nd = min(db_n[j], cap_db);
  const uint32_t* di = db_word + (size_t)j * cap_db;
  const float* dv = db_val + (size_t)j * cap_db;
  double acc = 0.0;
  if (SC == 3) {
    // KL: v1 = the query; every query word contributes, in query order
    for (int base = 0; base < nq; base += 64) {
      const int p = base + lane;
      const bool have = p < nq;
      const uint32_t id = have ? (stage ? s_id[p] : qi[p]) : 0xFFFFFFFFu;
      const float vi = have ? (stage ? s_val[p] : qv[p]) : 0.f;
      const int lo = lower_bound_u32(di, nd, id);
      const bool found = have && lo < nd && di[lo] == id;
      double term = 0.0;
      bool add = false;
      if (found) {
        const float wi = dv[lo];
        if (vi != 0.f && wi != 0.f) {
          term = (double)__fmul_rn(vi, logf(__fdiv_rn(vi, wi)));
          add = true;
        }
      } else if (have) {
        // v2 not exhausted (a larger id remains): added unconditionally inside the reference's loop; exhausted: the
        // tail loop skips zero values
        if (lo < nd || vi != 0.f) {
          term = (double)vi * ((double)logf(vi) - kLogEps);
          add = true;
        }
      }
      unsigned long long mask = __ballot(add);
      while (mask) {
        const int l = __builtin_ctzll(mask);
        mask &= mask - 1;
        acc += readlane_f64(term, l);
      }
    }
  } else {
    // symmetric scorers: only common words contribute, in ascending id order -- walk the database vector
    for (int base = 0; base < nd; base += 64) {
      const int p = base + lane;
      const bool have = p < nd;
      const uint32_t id = have ? di[p] : 0xFFFFFFFFu;
      const float wi = have ? dv[p] : 0.f;
      int lo;
      bool found;
      float vi = 0.f;
      if (stage) {
        lo = lower_bound_u32(s_id, nq, id);
        found = have && lo < nq && s_id[lo] == id;
        if (found) vi = s_val[lo];
      } else {
        lo = lower_bound_u32(qi, nq, id);
        found = have && lo < nq && qi[lo] == id;
        if (found) vi = qv[lo];
      }
      float term = 0.f;
      if (found) {
        if (SC == 0) term = __fsub_rn(__fsub_rn(fabsf(__fsub_rn(vi, wi)), fabsf(vi)), fabsf(wi));
        if (SC == 1 || SC == 5) term = __fmul_rn(vi, wi);
        if (SC == 2) {
          const float s = __fadd_rn(vi, wi);
          if (s != 0.f) term = __fdiv_rn(__fmul_rn(vi, wi), s); else found = false;
        }
        // correctly rounded float sqrt: a double sqrt rounded once more to float is exact for float inputs (53 >= 2 * 24 + 2);
        // the float intrinsic maps to the 1-ulp hardware instruction
        if (SC == 4) term = (float)__dsqrt_rn((double)__fmul_rn(vi, wi));
      }
      unsigned long long mask = __ballot(found);
      while (mask) {
        const int l = __builtin_ctzll(mask);
        mask &= mask - 1;
        acc += (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(term), l));
      }
    }
  }
  if (lane == 0) {
    double r = acc;
    if (SC == 0) r = -acc / 2.0;
    if (SC == 1) r = acc >= 1 ? 1.0 : 1.0 - __dsqrt_rn(1.0 - acc);
    if (SC == 2) r = 2. * acc;
    out[(size_t)q * n_db + j] = r;
  }
}

template <int SC>
gh_status launch_score(gh_ctx* ctx, const uint32_t* q_word, const float* q_val, const int32_t* q_n, int n_q, int cap_q,
                       const uint32_t* db_word, const float* db_val, const int32_t* db_n, int n_db, int cap_db,
                       double* out) {
  const int stage = cap_q <= 16384 ? 1 : 0;  // 8 B per word: up to 128 KB of the CU's 160 KB LDS
  const size_t lds = stage ? (size_t)cap_q * 8 : 0;
  if (lds > 48 * 1024)
    GH_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(bow_score_kernel<SC>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int q0 = 0; q0 < n_q; q0 += 65535) {
    const int nq = n_q - q0 < 65535 ? n_q - q0 : 65535;
    GH_LAUNCH(ctx, "bow_score", bow_score_kernel<SC>, dim3(gh_div_up(n_db, 4), nq), dim3(256), lds,
              q_word + (size_t)q0 * cap_q, q_val + (size_t)q0 * cap_q, q_n + q0, cap_q, db_word, db_val, db_n, cap_db,
              n_db, out + (size_t)q0 * n_db, stage);
  }
  return GH_OK;
}

}  // namespace

extern "C" gh_status gh_bow_score_dev(gh_ctx* ctx, int scoring, const uint32_t* q_word_dev, const float* q_val_dev,
                                      const int32_t* q_n_dev, int n_q, int cap_q, const uint32_t* db_word_dev,
                                      const float* db_val_dev, const int32_t* db_n_dev, int n_db, int cap_db,
                                      double* scores_dev) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, scoring >= 0 && scoring <= 5 && n_q >= 0 && n_db >= 0 && cap_q >= 0 && cap_db >= 0);
  if (n_q == 0 || n_db == 0) return GH_OK;
  GH_CHECK_ARG(ctx, q_n_dev && db_n_dev && scores_dev && (cap_q == 0 || (q_word_dev && q_val_dev)) &&
                        (cap_db == 0 || (db_word_dev && db_val_dev)));
#define GH_SCORE_CASE(SC)                                                                                             \
  case SC:                                                                                                            \
    return launch_score<SC>(ctx, q_word_dev, q_val_dev, q_n_dev, n_q, cap_q, db_word_dev, db_val_dev, db_n_dev, n_db, \
                            cap_db, scores_dev)
  switch (scoring) {
    GH_SCORE_CASE(0);
    GH_SCORE_CASE(1);
    GH_SCORE_CASE(2);
    GH_SCORE_CASE(3);
    GH_SCORE_CASE(4);
    GH_SCORE_CASE(5);
  }
#undef GH_SCORE_CASE
  return GH_ERR_ARG;
}

// One query against n_db database vectors given as CSR on the host (db_off[n_db + 1] offsets into db_word / db_val):
// what a loop detector holding std::map BowVectors calls.  Packs into the padded device layout, scores, downloads.
extern "C" gh_status gh_bow_score_host(gh_ctx* ctx, int scoring, const uint32_t* q_word, const float* q_val, int q_n,
                                       const uint32_t* db_word, const float* db_val, const int64_t* db_off, int n_db,
                                       double* scores) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, scoring >= 0 && scoring <= 5 && q_n >= 0 && n_db >= 0);
  if (n_db == 0) return GH_OK;
  GH_CHECK_ARG(ctx, db_off && scores && (q_n == 0 || (q_word && q_val)));
  int64_t cap = 1;
  for (int j = 0; j < n_db; ++j) {
    GH_CHECK_ARG(ctx, db_off[j + 1] >= db_off[j]);
    if (db_off[j + 1] - db_off[j] > cap) cap = db_off[j + 1] - db_off[j];
  }
  const int cap_db = (int)cap, cap_q = q_n > 0 ? q_n : 1;
  const size_t a = (((size_t)cap_q * 4) + 255) & ~(size_t)255, b = (((size_t)n_db * cap_db * 4) + 255) & ~(size_t)255;
  const size_t c = (((size_t)n_db * 4) + 255) & ~(size_t)255, d = (((size_t)n_db * 8) + 255) & ~(size_t)255;
  const size_t total = 2 * a + 256 + 2 * b + c + d;
  void *hp = nullptr, *dp = nullptr;
  GH_TRY(gh_pinned(ctx, total, &hp));
  GH_TRY(gh_scratch(ctx, total, &dp));
  uint8_t* h = (uint8_t*)hp;
  uint8_t* dv = (uint8_t*)dp;
  const size_t o_qw = 0, o_qv = a, o_qn = 2 * a, o_dw = 2 * a + 256, o_dv = o_dw + b, o_dn = o_dv + b, o_out = o_dn + c;
  if (q_n > 0) {
    memcpy(h + o_qw, q_word, (size_t)q_n * 4);
    memcpy(h + o_qv, q_val, (size_t)q_n * 4);
  }
  *(int32_t*)(h + o_qn) = q_n;
  for (int j = 0; j < n_db; ++j) {
    const int64_t n = db_off[j + 1] - db_off[j];
    if (n > 0) {
      memcpy(h + o_dw + (size_t)j * cap_db * 4, db_word + db_off[j], (size_t)n * 4);
      memcpy(h + o_dv + (size_t)j * cap_db * 4, db_val + db_off[j], (size_t)n * 4);
    }
    ((int32_t*)(h + o_dn))[j] = (int32_t)n;
  }
  GH_HIP(ctx, hipMemcpyAsync(dv, h, o_out, hipMemcpyHostToDevice, ctx->stream));
  GH_TRY(gh_bow_score_dev(ctx, scoring, (const uint32_t*)(dv + o_qw), (const float*)(dv + o_qv),
                          (const int32_t*)(dv + o_qn), 1, cap_q, (const uint32_t*)(dv + o_dw), (const float*)(dv + o_dv),
                          (const int32_t*)(dv + o_dn), n_db, cap_db, (double*)(dv + o_out)));
  GH_HIP(ctx, hipMemcpyAsync(h + o_out, dv + o_out, (size_t)n_db * 8, hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(scores, h + o_out, (size_t)n_db * 8);
  return GH_OK;
}
