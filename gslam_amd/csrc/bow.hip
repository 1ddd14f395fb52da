// Bag-of-words transform of GSLAM::Vocabulary on gfx950 (the step right after extraction).
//
// Bit-exact with the reference's own code (oracle/bow_oracle.c is pinned to it through oracle/_ref):
//   GSLAM/core/Vocabulary.h:1695-1736  per-feature greedy descent (children p*k+1 .. p*k+childNum, first strict
//                                      minimum, leaf = childNum == 0, node id at level L - levelsup)
//   GSLAM/core/Vocabulary.h:1558-1621  image transform (TF/TF_IDF accumulate, IDF/BINARY first weight, w <= 0 skipped)
//   GSLAM/core/Vocabulary.h:386-408    L1 / L2 normalisation, float /= double
//   GSLAM/core/Vocabulary.h:1843-1932  .gbow layout the vocabulary is created from
//
// CDNA4 mapping: words_kernel = one lane per descriptor, query in 8 VGPRs, children fetched as 2 x 16-byte loads
// (the tree's top levels stay L2 resident; k*L ~ 40-60 popcount distances per descriptor make this gather-bound).
// assemble_kernel = one workgroup per image: bitonic sort of the word ids in LDS, run-length -> value by repeated
// float addition (exactly what std::map += does), then ONE lane accumulates the norm in the reference's order
// (ascending word id, double) so the normalised floats are bit-identical.
#include "common.h"

namespace {

struct Node {
  uint32_t childNum;
  float weight;
};

// Descriptors are binary strings of 8 W8 bytes (GSLAM/core/Vocabulary.h:560-568: 32 -> hamming32, 64 -> hamming64, any
// other multiple of 8 -> hamming8x; all three are the popcount of the xor, 64 bits at a time).  W8 = 4 / 8 keep the query
// in registers; W8 = 0 is the run-time-length version (query re-read from memory for every child).
template <int W8>
__global__ __launch_bounds__(256) void bow_words_kernel(const Node* __restrict__ nodes, const uint2* __restrict__ ndesc, int k,
                                                        int L, int w8_rt, const uint2* __restrict__ desc,
                                                        const int32_t* __restrict__ counts, int cap, int levelsup,
                                                        uint32_t* __restrict__ word, float* __restrict__ weight,
                                                        uint32_t* __restrict__ node) {
  const int img = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = counts ? min(counts[img], cap) : cap;
  if (i >= cap) return;
  const size_t o = (size_t)img * cap + i;
  if (i >= n) {
    word[o] = 0xFFFFFFFFu;
    weight[o] = 0.f;
    node[o] = 0xFFFFFFFFu;
    return;
  }
  const int w8 = W8 ? W8 : w8_rt;
  uint2 q[W8 ? W8 : 1];
  if (W8) {
#pragma unroll
    for (int w = 0; w < W8; ++w) q[w] = desc[(size_t)W8 * o + w];
  }
  const int nid_level = L - levelsup;
  uint32_t final_id = 0, nid = 0;
  int level = 0;
  uint32_t cn = nodes[0].childNum;
  do {
    ++level;
    int best_d = 1 << 30;
    uint32_t best = final_id;
    const uint32_t first = final_id * (uint32_t)k + 1;
    for (uint32_t c = 0; c < cn; ++c) {
      const uint32_t id = first + c;
      int d = 0;
      if (W8) {
#pragma unroll
        for (int w = 0; w < W8; ++w) {
          const uint2 t = ndesc[(size_t)W8 * id + w];
          d += __popc(q[w].x ^ t.x) + __popc(q[w].y ^ t.y);
        }
      } else {
        for (int w = 0; w < w8; ++w) {
          const uint2 t = ndesc[(size_t)w8 * id + w], f = desc[(size_t)w8 * o + w];
          d += __popc(f.x ^ t.x) + __popc(f.y ^ t.y);
        }
      }
      if (d < best_d) {
        best_d = d;
        best = id;
      }
    }
    final_id = best;
    if (level == nid_level) nid = final_id;
    cn = nodes[final_id].childNum;
  } while (cn != 0);
  word[o] = final_id;
  weight[o] = nodes[final_id].weight;
  node[o] = nid_level <= 0 ? 0u : nid;
}

// Float descriptors (SIFT / SURF style vocabularies): the reference's l2generic (Vocabulary.h:550-560) -- squared L2,
// float accumulation in index order, one multiply and one add per component (this file is compiled with
// -ffp-contract=off: no fused multiply-add) -- which is what DistanceFactory::create selects for every dimension that is a
// multiple of 8 whatever ISA the host was built for (:569-578).  Start value FLT_MAX, first strict minimum (:1712-1725).
__global__ __launch_bounds__(256) void bow_words_f32_kernel(const Node* __restrict__ nodes, const float* __restrict__ ndesc, int k,
                                                            int L, int dims, const float* __restrict__ desc,
                                                            const int32_t* __restrict__ counts, int cap, int levelsup,
                                                            uint32_t* __restrict__ word, float* __restrict__ weight,
                                                            uint32_t* __restrict__ node) {
  const int img = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = counts ? min(counts[img], cap) : cap;
  if (i >= cap) return;
  const size_t o = (size_t)img * cap + i;
  if (i >= n) {
    word[o] = 0xFFFFFFFFu;
    weight[o] = 0.f;
    node[o] = 0xFFFFFFFFu;
    return;
  }
  const float* q = desc + o * (size_t)dims;
  const int nid_level = L - levelsup;
  uint32_t final_id = 0, nid = 0;
  int level = 0;
  uint32_t cn = nodes[0].childNum;
  do {
    ++level;
    float best_d = 3.402823466e+38f;
    uint32_t best = final_id;
    const uint32_t first = final_id * (uint32_t)k + 1;
    for (uint32_t c = 0; c < cn; ++c) {
      const uint32_t id = first + c;
      const float* t = ndesc + (size_t)id * dims;
      float sqd = 0.f;
      for (int e = 0; e < dims; ++e) {
        const float tmp = q[e] - t[e];
        sqd += tmp * tmp;
      }
      if (sqd < best_d) {
        best_d = sqd;
        best = id;
      }
    }
    if (best == final_id) break;  // no child compared below FLT_MAX (NaN / inf input): the reference would not terminate
    final_id = best;
    if (level == nid_level) nid = final_id;
    cn = nodes[final_id].childNum;
  } while (cn != 0);
  word[o] = final_id;
  weight[o] = nodes[final_id].weight;
  node[o] = nid_level <= 0 ? 0u : nid;
}

// one workgroup per image; dynamic LDS: uint32 keys[P] + float vals[P]
__global__ __launch_bounds__(256) void bow_assemble_kernel(const Node* __restrict__ nodes, int weighting, int scoring,
                                                           const uint32_t* __restrict__ word,
                                                           const float* __restrict__ weight, int cap, int P,
                                                           uint32_t* __restrict__ bow_word,
                                                           float* __restrict__ bow_val, int32_t* __restrict__ bow_n) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* keys = reinterpret_cast<uint32_t*>(smem);
  float* vals = reinterpret_cast<float*>(smem + (size_t)P * 4);
  __shared__ int s_wave[4];
  __shared__ int s_total;
  __shared__ double s_norm;
  const int img = blockIdx.x, tid = threadIdx.x;
  const size_t base = (size_t)img * cap;
  for (int i = tid; i < P; i += 256) keys[i] = (i < cap && weight[base + i] > 0.f) ? word[base + i] : 0xFFFFFFFFu;
  __syncthreads();
  // bitonic sort (ascending)
  for (int size = 2; size <= P; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (P >> 1); t += 256) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const uint32_t a = keys[lo], b = keys[hi];
        if ((a > b) == up) {
          keys[lo] = b;
          keys[hi] = a;
        }
      }
      __syncthreads();
    }
  // heads of runs -> output slot by block scan (P / 256 consecutive elements per thread)
  const int per = P >> 8 ? P >> 8 : 1;
  const int i0 = tid * per;
  int heads = 0;
  for (int e = 0; e < per; ++e) {
    const int i = i0 + e;
    if (i < P && keys[i] != 0xFFFFFFFFu && (i == 0 || keys[i - 1] != keys[i])) ++heads;
  }
  // exclusive scan of `heads` over the block
  int incl = heads;
  const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_wave[wv] = incl;
  __syncthreads();
  int off = 0;
  for (int w = 0; w < wv; ++w) off += s_wave[w];
  if (tid == 255) s_total = off + incl;
  int slot = off + incl - heads;
  __syncthreads();
  const int nb = s_total;
  for (int e = 0; e < per; ++e) {
    const int i = i0 + e;
    if (i < P && keys[i] != 0xFFFFFFFFu && (i == 0 || keys[i - 1] != keys[i])) {
      const uint32_t id = keys[i];
      int cnt = 1;
      while (i + cnt < P && keys[i + cnt] == id) ++cnt;
      const float w = nodes[id].weight;
      float v = w;
      if (weighting == 0 || weighting == 1)
        for (int c = 1; c < cnt; ++c) v = __fadd_rn(v, w);  // std::map value += w, in feature order
      bow_word[base + slot] = id;
      vals[slot] = v;
      ++slot;
    }
  }
  __syncthreads();
  const bool must = scoring != 5;
  if (tid == 0) {
    double norm = 0.0;
    if ((weighting == 0 || weighting == 1) && nb > 0 && !must) {
      norm = (double)nb;  // unnormalised TF: divide by the vector size
    } else if (must) {
      if (scoring == 1) {
        for (int i = 0; i < nb; ++i) norm += (double)__fmul_rn(vals[i], vals[i]);
        norm = sqrt(norm);
      } else {
        for (int i = 0; i < nb; ++i) norm += fabs((double)vals[i]);
      }
    }
    s_norm = norm;
    bow_n[img] = nb;
  }
  __syncthreads();
  const double norm = s_norm;
  for (int i = tid; i < nb; i += 256) {
    float v = vals[i];
    if (norm > 0.0) v = (float)((double)v / norm);
    bow_val[base + i] = v;
  }
  for (int i = nb + tid; i < cap; i += 256) {  // deterministic tail
    bow_word[base + i] = 0xFFFFFFFFu;
    bow_val[base + i] = 0.f;
  }
}

}  // namespace

struct gh_bow_vocab {
  gh_ctx* ctx;
  int k, L, weighting, scoring;
  uint32_t nnodes;
  Node* d_nodes;
  uint8_t* d_desc;
  int desc_bytes;  // a multiple of 8 (32 for ORB / BRIEF, 64 for the long binary descriptors); 4 * dims for float
  int is_float;    // descriptors are `desc_bytes / 4` floats compared with the squared L2 distance
};

extern "C" gh_status gh_bow_vocab_create(gh_ctx* ctx, int k, int L, int weighting, int scoring, uint32_t nnodes,
                                         const void* nodes, const uint8_t* node_desc, gh_bow_vocab** out) {
  return gh_bow_vocab_create_bytes(ctx, k, L, weighting, scoring, nnodes, nodes, node_desc, 32, out);
}

extern "C" gh_status gh_bow_vocab_create_f32(gh_ctx* ctx, int k, int L, int weighting, int scoring, uint32_t nnodes,
                                             const void* nodes, const float* node_desc, int dims, gh_bow_vocab** out) {
  if (!ctx || !out) return GH_ERR_ARG;
  {
    GH_ENTER(ctx);
    GH_CHECK_ARG(ctx, dims >= 8 && dims <= 4096 && dims % 8 == 0);
  }
  const gh_status st = gh_bow_vocab_create_bytes(ctx, k, L, weighting, scoring, nnodes, nodes, (const uint8_t*)node_desc, 4 * dims, out);
  if (st == GH_OK) (*out)->is_float = 1;
  return st;
}

extern "C" gh_status gh_bow_vocab_create_bytes(gh_ctx* ctx, int k, int L, int weighting, int scoring, uint32_t nnodes,
                                               const void* nodes, const uint8_t* node_desc, int desc_bytes, gh_bow_vocab** out) {
  if (!ctx || !out) return GH_ERR_ARG;
  GH_ENTER(ctx);
  *out = nullptr;
  GH_CHECK_ARG(ctx, k >= 2 && L >= 1 && nnodes >= 1 && nodes && node_desc);
  GH_CHECK_ARG(ctx, desc_bytes >= 8 && desc_bytes <= 16384 && desc_bytes % 8 == 0);
  GH_CHECK_ARG(ctx, weighting >= 0 && weighting <= 3 && scoring >= 0 && scoring <= 5);
  const Node* hn = (const Node*)nodes;
  for (uint32_t i = 0; i < nnodes; ++i)  // every child index must exist (the descent never checks bounds)
    GH_CHECK_ARG(ctx, hn[i].childNum <= (uint32_t)k && (hn[i].childNum == 0 || (uint64_t)i * k + hn[i].childNum < nnodes));
  gh_bow_vocab* v = new (std::nothrow) gh_bow_vocab();
  if (!v) return GH_ERR_NOMEM;
  *v = gh_bow_vocab{ctx, k, L, weighting, scoring, nnodes, nullptr, nullptr, desc_bytes, 0};
  gh_status st = gh_dev_alloc(ctx, (size_t)nnodes * sizeof(Node), (void**)&v->d_nodes);
  if (st == GH_OK) st = gh_dev_alloc(ctx, (size_t)nnodes * desc_bytes, (void**)&v->d_desc);
  if (st == GH_OK) st = gh_dev_upload(ctx, v->d_nodes, nodes, (size_t)nnodes * sizeof(Node));
  if (st == GH_OK) st = gh_dev_upload(ctx, v->d_desc, node_desc, (size_t)nnodes * desc_bytes);
  if (st != GH_OK) {
    if (v->d_nodes) hipFree(v->d_nodes);
    if (v->d_desc) hipFree(v->d_desc);
    delete v;
    return st;
  }
  *out = v;
  return GH_OK;
}

extern "C" void gh_bow_vocab_destroy(gh_bow_vocab* v) {
  if (!v) return;
  GH_ENTER(v->ctx);
  hipStreamSynchronize(v->ctx->stream);
  hipFree(v->d_nodes);
  hipFree(v->d_desc);
  delete v;
}

extern "C" gh_status gh_bow_transform_dev(gh_bow_vocab* v, const uint8_t* desc_dev, const int32_t* counts_dev, int cap,
                                          int n_images, int levelsup, uint32_t* word_dev, float* weight_dev,
                                          uint32_t* node_dev, uint32_t* bow_word_dev, float* bow_val_dev,
                                          int32_t* bow_n_dev) {
  if (!v) return GH_ERR_ARG;
  gh_ctx* ctx = v->ctx;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, cap >= 0 && cap <= 16384 && n_images >= 0 && n_images <= 65535);
  if (cap == 0 || n_images == 0) return GH_OK;
  GH_CHECK_ARG(ctx, desc_dev && word_dev && weight_dev && node_dev && bow_word_dev && bow_val_dev && bow_n_dev);
  GH_CHECK_ARG(ctx, ((uintptr_t)desc_dev & 7) == 0);
  const int w8 = v->desc_bytes / 8;
  const dim3 wgrid(gh_div_up(cap, 256), n_images);
  if (v->is_float)
    GH_LAUNCH(ctx, "bow_words", bow_words_f32_kernel, wgrid, dim3(256), 0, v->d_nodes, (const float*)v->d_desc, v->k, v->L,
              v->desc_bytes / 4, (const float*)desc_dev, counts_dev, cap, levelsup, word_dev, weight_dev, node_dev);
  else if (w8 == 4)
    GH_LAUNCH(ctx, "bow_words", bow_words_kernel<4>, wgrid, dim3(256), 0, v->d_nodes, (const uint2*)v->d_desc, v->k, v->L, w8,
              (const uint2*)desc_dev, counts_dev, cap, levelsup, word_dev, weight_dev, node_dev);
  else if (w8 == 8)
    GH_LAUNCH(ctx, "bow_words", bow_words_kernel<8>, wgrid, dim3(256), 0, v->d_nodes, (const uint2*)v->d_desc, v->k, v->L, w8,
              (const uint2*)desc_dev, counts_dev, cap, levelsup, word_dev, weight_dev, node_dev);
  else
    GH_LAUNCH(ctx, "bow_words", bow_words_kernel<0>, wgrid, dim3(256), 0, v->d_nodes, (const uint2*)v->d_desc, v->k, v->L, w8,
              (const uint2*)desc_dev, counts_dev, cap, levelsup, word_dev, weight_dev, node_dev);
  int P = 256;
  while (P < cap) P <<= 1;
  if ((size_t)P * 8 > 48 * 1024) {
    // more than the default dynamic-LDS allowance: a gfx950 workgroup may take all 160 KB of its CU (P = 16384: 128 KB)
    GH_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(bow_assemble_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, P * 8));
  }
  GH_LAUNCH(ctx, "bow_assemble", bow_assemble_kernel, dim3(n_images), dim3(256), (size_t)P * 8, v->d_nodes, v->weighting,
            v->scoring, word_dev, weight_dev, cap, P, bow_word_dev, bow_val_dev, bow_n_dev);
  return GH_OK;
}

extern "C" gh_status gh_bow_transform_host(gh_bow_vocab* v, const uint8_t* desc, int n, int levelsup, uint32_t* word,
                                           float* weight, uint32_t* node, uint32_t* bow_word, float* bow_val,
                                           int32_t* bow_n) {
  if (!v) return GH_ERR_ARG;
  gh_ctx* ctx = v->ctx;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, n >= 0 && n <= 16384 && bow_n);
  *bow_n = 0;
  if (n == 0) return GH_OK;
  GH_CHECK_ARG(ctx, desc && word && weight && node && bow_word && bow_val);
  const size_t a = ((size_t)n * v->desc_bytes + 255) & ~(size_t)255, b = ((size_t)n * 4 + 255) & ~(size_t)255;
  const size_t total = a + 5 * b + 256;
  void *s = nullptr, *hs = nullptr;
  GH_TRY(gh_scratch(ctx, total, &s));
  GH_TRY(gh_pinned(ctx, total, &hs));
  uint8_t *p = (uint8_t*)s, *hp = (uint8_t*)hs;
  uint8_t* d_desc = p;
  uint32_t* d_word = (uint32_t*)(p + a);
  float* d_weight = (float*)(p + a + b);
  uint32_t* d_node = (uint32_t*)(p + a + 2 * b);
  uint32_t* d_bw = (uint32_t*)(p + a + 3 * b);
  float* d_bv = (float*)(p + a + 4 * b);
  int32_t* d_n = (int32_t*)(p + a + 5 * b);
  // one DMA up (descriptors) and one down (word | weight | node | bow_word | bow_val | bow_n) through the context's pinned
  // block, as gh_bf_match_host does: a per-frame caller (Vocabulary::transform of one image) paid seven pageable copies
  // of ~30-50 us each for two kernels of ~20 us
  memcpy(hp, desc, (size_t)n * v->desc_bytes);
  GH_HIP(ctx, hipMemcpyAsync(d_desc, hp, (size_t)n * v->desc_bytes, hipMemcpyHostToDevice, ctx->stream));
  GH_TRY(gh_bow_transform_dev(v, d_desc, nullptr, n, 1, levelsup, d_word, d_weight, d_node, d_bw, d_bv, d_n));
  GH_HIP(ctx, hipMemcpyAsync(hp + a, p + a, 5 * b + 4, hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const size_t nb = (size_t)n * 4;
  memcpy(word, hp + a, nb);
  memcpy(weight, hp + a + b, nb);
  memcpy(node, hp + a + 2 * b, nb);
  memcpy(bow_word, hp + a + 3 * b, nb);
  memcpy(bow_val, hp + a + 4 * b, nb);
  memcpy(bow_n, hp + a + 5 * b, 4);
  return GH_OK;
}
