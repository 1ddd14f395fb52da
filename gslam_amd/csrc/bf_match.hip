// Brute-force 256-bit Hamming matcher for gfx950.
//
// Semantics (bit-exact with oracle/bf_oracle.c):
//   distance  = GSLAM/core/Vocabulary.h:485-491 (hamming32: XOR + popcount over 4 x u64)
//   selection = GSLAM/core/Vocabulary.h:1712-1725 (strict '<' => lowest train index wins ties)
//
// Mapping to CDNA4: this is popcount work, bound by VALU issue (8 v_xor_b32 + 8 accumulating
// v_bcnt_u32_b32 per pair), not by HBM and not MFMA-shaped.  One wave owns 64*QPT query rows
// held in VGPRs for the whole kernel; the train descriptor is wave-uniform, so it is fetched
// with scalar loads (s_load_dwordx8 = one descriptor) and XOR-ed straight from SGPRs - no LDS,
// no barrier.  Best/second-best are tracked on a packed key (dist << 16 | train_index) so
// the first-minimum tie rule falls out of an unsigned min: v_lshl_or + v_med3_u32 + v_min_u32.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int kQPT = 2;            // query rows per lane
constexpr int kWaveQueries = 64 * kQPT;

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

struct QueryRegs {
  uint32_t w[8];
};

// key = dist << 16 | train index (train index is wave-uniform -> SGPR operand).
__device__ __forceinline__ uint32_t make_key(uint32_t d, uint32_t j) {
  uint32_t r;
  asm("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(r) : "v"(d), "s"(j));
  return r;
}

__device__ __forceinline__ uint32_t dist256(const QueryRegs& q, const uint32_t (&t)[8]) {
  // v_bcnt_u32_b32 D = popcount(S0) + S1: keep the accumulate form (the compiler otherwise
  // splits the sum into 8 independent bcnt + 3 v_add3_u32).
  uint32_t d;
  {
    uint32_t x = q.w[0] ^ t[0];
    asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(d) : "v"(x));
  }
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    uint32_t x = q.w[k] ^ t[k];
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d) : "v"(x));
  }
  return d;
}

// q_words / t_words: descriptor matrices as dwords (8 per row).  One wave per 64*QPT queries.
__device__ __forceinline__ void bf_wave(const uint32_t* __restrict__ q_words, int nq,
                                        const uint32_t* __restrict__ t_words, int nt, int out_rows,
                                        int32_t* __restrict__ idx1, uint16_t* __restrict__ d1,
                                        uint16_t* __restrict__ d2) {
  const int lane = threadIdx.x;
  const int q0 = blockIdx.x * kWaveQueries;
  QueryRegs q[kQPT];
#pragma unroll
  for (int r = 0; r < kQPT; ++r) {
    int qi = q0 + r * 64 + lane;
    int qc = qi < nq ? qi : (nq > 0 ? nq - 1 : 0);
    if (nq > 0) {
      const uint4* p = reinterpret_cast<const uint4*>(q_words + (size_t)qc * 8);
      uint4 a = p[0], b = p[1];
      q[r].w[0] = a.x; q[r].w[1] = a.y; q[r].w[2] = a.z; q[r].w[3] = a.w;
      q[r].w[4] = b.x; q[r].w[5] = b.y; q[r].w[6] = b.z; q[r].w[7] = b.w;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) q[r].w[k] = 0;
    }
  }
  uint32_t best[kQPT], second[kQPT];
#pragma unroll
  for (int r = 0; r < kQPT; ++r) best[r] = second[r] = 0xFFFFFFFFu;

  int j = 0;
  // Main loop: 4 train descriptors (32 SGPRs) per trip; t_words + j*8 is wave-uniform.
  for (; j + 4 <= nt; j += 4) {
    uint32_t t[4][8];
    const uint32_t* tp = t_words + (size_t)j * 8;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 8; ++k) t[u][k] = tp[u * 8 + k];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int r = 0; r < kQPT; ++r) {
        uint32_t key = make_key(dist256(q[r], t[u]), (uint32_t)(j + u));
        second[r] = umed3(key, best[r], second[r]);
        best[r] = min(best[r], key);
      }
    }
  }
  for (; j < nt; ++j) {
    uint32_t t[8];
    const uint32_t* tp = t_words + (size_t)j * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = tp[k];
#pragma unroll
    for (int r = 0; r < kQPT; ++r) {
      uint32_t key = make_key(dist256(q[r], t), (uint32_t)j);
      second[r] = umed3(key, best[r], second[r]);
      best[r] = min(best[r], key);
    }
  }
#pragma unroll
  for (int r = 0; r < kQPT; ++r) {
    int qi = q0 + r * 64 + lane;
    if (qi < out_rows) {
      bool valid = qi < nq;
      uint32_t b = valid ? best[r] : 0xFFFFFFFFu, s = valid ? second[r] : 0xFFFFFFFFu;
      idx1[qi] = (b == 0xFFFFFFFFu) ? -1 : (int32_t)(b & 0xFFFFu);
      d1[qi] = (uint16_t)(b >> 16);
      d2[qi] = (uint16_t)(s >> 16);
    }
  }
}

__global__ __launch_bounds__(64) void bf_match_single_kernel(const uint32_t* __restrict__ q, int nq,
                                                             const uint32_t* __restrict__ t, int nt,
                                                             int32_t* __restrict__ idx1, uint16_t* __restrict__ d1,
                                                             uint16_t* __restrict__ d2) {
  bf_wave(q, nq, t, nt, nq, idx1, d1, d2);
}

// One frame pair alone (the per-frame call of a tracking front end) leaves the machine empty with one wave per 64 * kQPT
// queries: the kernel is then a latency chain of nt x kQPT distance evaluations per wave (~100 us at 1000 x 1000).  Here a
// workgroup takes 64 queries and its S waves split the TRAIN rows; the partial (best, second) keys meet in LDS.  The
// keys carry the global train index, so the merged minimum / second minimum are the ones of the single sweep, bit for bit.
__global__ __launch_bounds__(1024) void bf_match_split_kernel(const uint32_t* __restrict__ q_words, int nq,
                                                              const uint32_t* __restrict__ t_words, int nt, int chunk,
                                                              int32_t* __restrict__ idx1, uint16_t* __restrict__ d1,
                                                              uint16_t* __restrict__ d2) {
  __shared__ uint32_t sb[16][64], ss[16][64];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), S = blockDim.x >> 6;
  const int qi = blockIdx.x * 64 + lane;
  const int qc = qi < nq ? qi : nq - 1;
  QueryRegs q;
  {
    const uint4* p = reinterpret_cast<const uint4*>(q_words + (size_t)qc * 8);
    const uint4 a = p[0], b = p[1];
    q.w[0] = a.x; q.w[1] = a.y; q.w[2] = a.z; q.w[3] = a.w;
    q.w[4] = b.x; q.w[5] = b.y; q.w[6] = b.z; q.w[7] = b.w;
  }
  uint32_t best = 0xFFFFFFFFu, second = 0xFFFFFFFFu;
  int j = w * chunk;
  const int j1 = min(nt, j + chunk);
  for (; j + 4 <= j1; j += 4) {
    uint32_t t[4][8];
    const uint32_t* tp = t_words + (size_t)j * 8;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 8; ++k) t[u][k] = tp[u * 8 + k];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t key = make_key(dist256(q, t[u]), (uint32_t)(j + u));
      second = umed3(key, best, second);
      best = min(best, key);
    }
  }
  for (; j < j1; ++j) {
    uint32_t t[8];
    const uint32_t* tp = t_words + (size_t)j * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = tp[k];
    const uint32_t key = make_key(dist256(q, t), (uint32_t)j);
    second = umed3(key, best, second);
    best = min(best, key);
  }
  sb[w][lane] = best;
  ss[w][lane] = second;
  __syncthreads();
  if (w != 0 || qi >= nq) return;
  for (int s2 = 1; s2 < S; ++s2) {
    const uint32_t kb = sb[s2][lane], ks = ss[s2][lane];
    second = umed3(kb, best, second);
    best = min(best, kb);
    second = umed3(ks, best, second);
    best = min(best, ks);
  }
  idx1[qi] = (best == 0xFFFFFFFFu) ? -1 : (int32_t)(best & 0xFFFFu);
  d1[qi] = (uint16_t)(best >> 16);
  d2[qi] = (uint16_t)(second >> 16);
}

__global__ __launch_bounds__(64) void bf_match_pairs_kernel(const uint32_t* __restrict__ desc,
                                                            const int32_t* __restrict__ counts, int cap,
                                                            const int32_t* __restrict__ pair_q,
                                                            const int32_t* __restrict__ pair_t,
                                                            int32_t* __restrict__ idx1, uint16_t* __restrict__ d1,
                                                            uint16_t* __restrict__ d2) {
  const int p = blockIdx.y;
  const int fq = pair_q[p], ft = pair_t[p];
  int nq = counts[fq], nt = counts[ft];
  nq = nq < cap ? nq : cap;
  nt = nt < cap ? nt : cap;
  const size_t o = (size_t)p * cap;
  bf_wave(desc + (size_t)fq * cap * 8, nq, desc + (size_t)ft * cap * 8, nt, cap, idx1 + o, d1 + o, d2 + o);
}

// Row-band restricted variant (stereo left-right matching, SURVEY.md 8e "C3"): train row j is a candidate of query
// i only if |y_i - y_j| <= band_i with band_i = size_i * band_per_size (size = 31 * scale^octave, so the band grows
// with the pyramid level as in ORB-SLAM's stereo matcher).  Same first-minimum / second-minimum rules.
__global__ __launch_bounds__(64) void bf_match_band_pairs_kernel(const uint32_t* __restrict__ desc,
                                                                 const gh_keypoint* __restrict__ kps,
                                                                 const int32_t* __restrict__ counts, int cap,
                                                                 const int32_t* __restrict__ pair_q,
                                                                 const int32_t* __restrict__ pair_t,
                                                                 float band_per_size, int32_t* __restrict__ idx1,
                                                                 uint16_t* __restrict__ d1, uint16_t* __restrict__ d2) {
  const int p = blockIdx.y;
  const int fq = pair_q[p], ft = pair_t[p];
  int nq = counts[fq], nt = counts[ft];
  nq = nq < cap ? nq : cap;
  nt = nt < cap ? nt : cap;
  const int qi = blockIdx.x * 64 + threadIdx.x;
  if (qi >= cap) return;
  const size_t o = (size_t)p * cap + qi;
  if (qi >= nq) {
    idx1[o] = -1;
    d1[o] = 65535;
    d2[o] = 65535;
    return;
  }
  const uint32_t* qw = desc + ((size_t)fq * cap + qi) * 8;
  QueryRegs q;
#pragma unroll
  for (int k = 0; k < 8; ++k) q.w[k] = qw[k];
  const gh_keypoint kq = kps[(size_t)fq * cap + qi];
  const float yq = kq.y, band = __fmul_rn(kq.size, band_per_size);
  uint32_t best = 0xFFFFFFFFu, second = 0xFFFFFFFFu;
  const uint32_t* tw = desc + (size_t)ft * cap * 8;
  const gh_keypoint* tk = kps + (size_t)ft * cap;
  for (int j = 0; j < nt; ++j) {
    uint32_t t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = tw[(size_t)j * 8 + k];
    const float dy = fabsf(__fsub_rn(yq, tk[j].y));
    uint32_t key = make_key(dist256(q, t), (uint32_t)j);
    key = dy <= band ? key : 0xFFFFFFFFu;
    second = umed3(key, best, second);
    best = min(best, key);
  }
  idx1[o] = best == 0xFFFFFFFFu ? -1 : (int32_t)(best & 0xFFFFu);
  d1[o] = (uint16_t)(best >> 16);
  d2[o] = (uint16_t)(second >> 16);
}

__global__ void match_mask_kernel(const int32_t* __restrict__ idx1, const uint16_t* __restrict__ d1,
                                  const uint16_t* __restrict__ d2, int nq, const int32_t* __restrict__ back, int nt,
                                  int max_dist, int ratio_num, int ratio_den, int cross_check,
                                  uint8_t* __restrict__ keep) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  int j = idx1[i];
  bool ok = j >= 0 && (int)d1[i] <= max_dist;
  if (ok && ratio_num > 0) ok = (int)d1[i] * ratio_den < ratio_num * (int)d2[i];
  if (ok && cross_check) ok = j < nt && back[j] == i;
  keep[i] = ok ? 1 : 0;
}

// VALU ceiling probe: 16 dependent-free xor+bcnt chains per lane, no memory traffic.
__global__ __launch_bounds__(256) void valu_probe_kernel(uint32_t* out, int iters, uint32_t seed) {
  uint32_t q[8], t[8], acc[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    q[k] = seed * (k + 1) + threadIdx.x;
    t[k] = seed ^ (0x9E3779B9u * (k + 3));
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        uint32_t x = q[k] ^ t[k];
        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[u]) : "v"(x));
      }
      // perturb the uniform operand so the compiler cannot hoist the chain
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("" : "+s"(t[k]));
    }
  }
  if ((acc[0] + acc[1] + acc[2] + acc[3]) == 0xFFFFFFFFu) out[0] = 1;
}

// Train sets beyond 65535 rows (a frame against a local map): the 16-bit index field of the search key holds one CHUNK;
// chunks are matched one after the other and folded into the running (best, index, second best) with the same strict '<':
// an equal distance in a later chunk never displaces the earlier index (GSLAM/core/Vocabulary.h:1712-1725) and becomes the
// second best, exactly as in one sweep over all rows.
__global__ void bf_merge_chunk_kernel(int nq, int32_t* __restrict__ idx1, uint16_t* __restrict__ d1, uint16_t* __restrict__ d2,
                                      const int32_t* __restrict__ cidx, const uint16_t* __restrict__ cd1,
                                      const uint16_t* __restrict__ cd2, int base) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq || cidx[i] < 0) return;
  const unsigned b1 = d1[i], b2 = d2[i], c1 = cd1[i], c2 = cd2[i];
  const bool had = idx1[i] >= 0;
  if (!had || c1 < b1) {
    idx1[i] = base + cidx[i];
    d1[i] = (uint16_t)c1;
    d2[i] = (uint16_t)(!had ? c2 : (b1 < c2 ? b1 : c2));
  } else {
    d2[i] = (uint16_t)(c1 < b2 ? c1 : b2);
  }
}

}  // namespace

static gh_status bf_match_chunk(gh_ctx* ctx, const uint8_t* q_dev, int nq, const uint8_t* t_dev, int nt, int32_t* idx1_dev,
                                uint16_t* d1_dev, uint16_t* d2_dev);

// bytes of device memory bf_match_any wants for a train set of nt rows (0 up to 65535 rows)
static size_t bf_chunk_tmp_bytes(int nq, int nt) {
  return nt <= 65535 ? 0 : ((((size_t)nq * 4) + 255) & ~(size_t)255) + 2 * ((((size_t)nq * 2) + 255) & ~(size_t)255);
}
static gh_status bf_match_any(gh_ctx* ctx, const uint8_t* q_dev, int nq, const uint8_t* t_dev, int nt, int32_t* idx1_dev,
                              uint16_t* d1_dev, uint16_t* d2_dev, void* tmp);

extern "C" gh_status gh_bf_match_dev(gh_ctx* ctx, const uint8_t* q_dev, int nq, const uint8_t* t_dev, int nt,
                                     int32_t* idx1_dev, uint16_t* d1_dev, uint16_t* d2_dev) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, nq >= 0 && nt >= 0);
  if (nq == 0) return GH_OK;
  GH_CHECK_ARG(ctx, q_dev && idx1_dev && d1_dev && d2_dev && (nt == 0 || t_dev));
  GH_CHECK_ARG(ctx, ((uintptr_t)q_dev & 15) == 0 && ((uintptr_t)t_dev & 3) == 0);
  void* tmp = nullptr;
  if (nt > 65535) GH_TRY(gh_scratch(ctx, bf_chunk_tmp_bytes(nq, nt), &tmp));
  return bf_match_any(ctx, q_dev, nq, t_dev, nt, idx1_dev, d1_dev, d2_dev, tmp);
}

static gh_status bf_match_any(gh_ctx* ctx, const uint8_t* q_dev, int nq, const uint8_t* t_dev, int nt, int32_t* idx1_dev,
                              uint16_t* d1_dev, uint16_t* d2_dev, void* tmp) {
  if (nt <= 65535) return bf_match_chunk(ctx, q_dev, nq, t_dev, nt, idx1_dev, d1_dev, d2_dev);
  // chunks of 65532 rows (a multiple of 4: the split kernel's row quads), results of chunk 0 straight into the outputs
  constexpr int kChunk = 65532;
  const size_t ib = (((size_t)nq * 4) + 255) & ~(size_t)255, db = (((size_t)nq * 2) + 255) & ~(size_t)255;
  int32_t* cidx = (int32_t*)tmp;
  uint16_t* cd1 = (uint16_t*)((uint8_t*)tmp + ib);
  uint16_t* cd2 = (uint16_t*)((uint8_t*)tmp + ib + db);
  for (int base = 0; base < nt; base += kChunk) {
    const int n = nt - base < kChunk ? nt - base : kChunk;
    const uint8_t* tc = t_dev + (size_t)base * 32;
    if (base == 0) {
      GH_TRY(bf_match_chunk(ctx, q_dev, nq, tc, n, idx1_dev, d1_dev, d2_dev));
    } else {
      GH_TRY(bf_match_chunk(ctx, q_dev, nq, tc, n, cidx, cd1, cd2));
      GH_LAUNCH(ctx, "bf_merge_chunk", bf_merge_chunk_kernel, dim3(gh_div_up(nq, 256)), dim3(256), 0, nq, idx1_dev, d1_dev, d2_dev,
                cidx, cd1, cd2, base);
    }
  }
  return GH_OK;
}

static gh_status bf_match_chunk(gh_ctx* ctx, const uint8_t* q_dev, int nq, const uint8_t* t_dev, int nt, int32_t* idx1_dev,
                                uint16_t* d1_dev, uint16_t* d2_dev) {
  // few queries: split the train rows over the waves of a workgroup (latency); many: one wave per 64 * kQPT queries
  static const bool split_env = [] {
    const char* e = getenv("GSLAM_HIP_BF_SPLIT");  // "0": always the one-wave sweep (A/B measurements)
    return !(e && e[0] == '0');
  }();
  const int S = nt / 64 < 1 ? 1 : (nt / 64 > 16 ? 16 : nt / 64);
  if (split_env && S > 1 && (long long)gh_div_up(nq, 64) * S <= 4096) {
    const int chunk = (gh_div_up(nt, S) + 3) & ~3;
    GH_LAUNCH(ctx, "bf_match_split", bf_match_split_kernel, dim3(gh_div_up(nq, 64)), dim3(64 * S), 0, (const uint32_t*)q_dev, nq,
              (const uint32_t*)t_dev, nt, chunk, idx1_dev, d1_dev, d2_dev);
    return GH_OK;
  }
  dim3 grid(gh_div_up(nq, kWaveQueries));
  GH_LAUNCH(ctx, "bf_match", bf_match_single_kernel, grid, dim3(64), 0, (const uint32_t*)q_dev, nq,
            (const uint32_t*)t_dev, nt, idx1_dev, d1_dev, d2_dev);
  return GH_OK;
}

extern "C" gh_status gh_bf_match_host(gh_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx1,
                                      uint16_t* d1, uint16_t* d2) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, nq >= 0 && nt >= 0);
  if (nq == 0) return GH_OK;
  size_t qb = (size_t)nq * 32, tb = (size_t)nt * 32;
  size_t off_t = (qb + 255) & ~(size_t)255;
  size_t off_i = off_t + ((tb + 255) & ~(size_t)255);
  size_t off_d1 = off_i + (((size_t)nq * 4 + 255) & ~(size_t)255);
  size_t off_d2 = off_d1 + (((size_t)nq * 2 + 255) & ~(size_t)255);
  size_t total = off_d2 + (size_t)nq * 2;
  const size_t off_tmp = (total + 255) & ~(size_t)255;  // chunk results of a train set beyond 65535 rows (device side only)
  const size_t dev_total = off_tmp + bf_chunk_tmp_bytes(nq, nt);
  void *base = nullptr, *hbase = nullptr;
  GH_TRY(gh_scratch(ctx, dev_total, &base));
  GH_TRY(gh_pinned(ctx, total, &hbase));
  uint8_t *b = (uint8_t*)base, *hb = (uint8_t*)hbase;
  // one DMA up (q | t) and one down (idx1 | d1 | d2) through the context's pinned block: a pageable copy per array
  // costs ~50 us each on this stack, the whole 2000 x 2000 match kernel ~5 us
  memcpy(hb, q, qb);
  if (tb) memcpy(hb + off_t, t, tb);
  GH_HIP(ctx, hipMemcpyAsync(b, hb, off_t + tb, hipMemcpyHostToDevice, ctx->stream));
  GH_TRY(bf_match_any(ctx, b, nq, b + off_t, nt, (int32_t*)(b + off_i), (uint16_t*)(b + off_d1), (uint16_t*)(b + off_d2),
                      b + off_tmp));
  GH_HIP(ctx, hipMemcpyAsync(hb + off_i, b + off_i, total - off_i, hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(idx1, hb + off_i, (size_t)nq * 4);
  memcpy(d1, hb + off_d1, (size_t)nq * 2);
  memcpy(d2, hb + off_d2, (size_t)nq * 2);
  return GH_OK;
}

extern "C" gh_status gh_bf_match_pairs_popc_dev(gh_ctx* ctx, const uint8_t* desc_dev, const int32_t* counts_dev, int cap,
                                                const int32_t* pair_q_dev, const int32_t* pair_t_dev, int npairs,
                                                int32_t* idx1_dev, uint16_t* d1_dev, uint16_t* d2_dev) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, cap >= 0 && cap <= 65535 && npairs >= 0);
  if (npairs == 0 || cap == 0) return GH_OK;
  GH_CHECK_ARG(ctx, desc_dev && counts_dev && pair_q_dev && pair_t_dev && idx1_dev && d1_dev && d2_dev);
  GH_CHECK_ARG(ctx, ((uintptr_t)desc_dev & 15) == 0);
  const int qblocks = gh_div_up(cap, kWaveQueries);
  // grid.y is limited to 65535: chunk the pair list.
  for (int p0 = 0; p0 < npairs; p0 += 65535) {
    int np = npairs - p0 < 65535 ? npairs - p0 : 65535;
    dim3 grid(qblocks, np);
    size_t o = (size_t)p0 * cap;
    GH_LAUNCH(ctx, "bf_match_pairs", bf_match_pairs_kernel, grid, dim3(64), 0, (const uint32_t*)desc_dev, counts_dev,
              cap, pair_q_dev + p0, pair_t_dev + p0, idx1_dev + o, d1_dev + o, d2_dev + o);
  }
  return GH_OK;
}

// The batched entry the plugins and bench.py call.  Both kernels return the same bits (tests/test_bf_gpu.py,
// __graft_entry__.smoke): the popcount kernel is the contract formulation of north_star and serves small batches; a batch
// with enough pair work to fill the chip goes through the exact MFMA formulation (bf_match_mfma.hip), 2.8x faster and on
// the otherwise idle matrix pipe.  GSLAM_HIP_BF_MFMA = 0 / 1 forces one or the other (A/B measurements).
extern "C" gh_status gh_bf_match_pairs_dev(gh_ctx* ctx, const uint8_t* desc_dev, const int32_t* counts_dev, int cap,
                                           const int32_t* pair_q_dev, const int32_t* pair_t_dev, int npairs,
                                           int32_t* idx1_dev, uint16_t* d1_dev, uint16_t* d2_dev) {
  if (!ctx) return GH_ERR_ARG;
  static const int force = [] {
    const char* e = getenv("GSLAM_HIP_BF_MFMA");
    return e && (e[0] == '0' || e[0] == '1') ? e[0] - '0' : -1;
  }();
  // a workgroup of the MFMA kernel takes 512 queries of one frame pair and expands the whole train set for them: it pays
  // from a few hundred rows per frame and enough frame pairs to occupy the CUs
  const bool big = cap >= 512 && cap <= 65535 && (long long)npairs * cap * cap >= (1LL << 27);
  const bool mfma = force >= 0 ? force == 1 : big;
  if (mfma && ((uintptr_t)desc_dev & 7) == 0)
    return gh_bf_match_pairs_mfma_dev(ctx, desc_dev, counts_dev, cap, pair_q_dev, pair_t_dev, npairs, idx1_dev, d1_dev, d2_dev);
  return gh_bf_match_pairs_popc_dev(ctx, desc_dev, counts_dev, cap, pair_q_dev, pair_t_dev, npairs, idx1_dev, d1_dev, d2_dev);
}

// ---------------------------------------------------------------------------------------------------------------------
// Descriptors of ANY width that is a multiple of 8 bytes: GSLAM/core/Vocabulary.h:493-513 (hamming64: 64-byte rows -- BRISK /
// FREAK sized; hamming8x: bytes / 8 uint64_t words) with the same first-minimum rule (:1712-1725).  The reference picks the
// distance by the descriptor's width (DistanceFactory::create, :565-567); so do these entries: 32 bytes run the kernels above.
// One query row per lane (64 per wave), the train row wave-uniform through scalar loads as above; W = 16 dwords holds the
// query in registers, other widths read it from an LDS copy of the wave's 64 rows.  Keys as above (distance <= 2048 fits).
namespace {
struct BytesArgs {
  const uint32_t* q;   // single: query rows; pairs: the descriptor batch
  const uint32_t* t;   // single: train rows
  const int32_t *counts, *pair_q, *pair_t;  // pairs (null = single)
  int nq, nt, cap, words;
  int32_t* idx1;
  uint16_t *d1, *d2;
};

template <int W>
__global__ __launch_bounds__(64) void bf_match_bytes_kernel(BytesArgs a) {
  extern __shared__ uint32_t qs[];  // W == 0: [words][64] query words of the wave
  const int lane = threadIdx.x;
  const int words = W ? W : a.words;
  const uint32_t *qrows = a.q, *trows = a.t;
  int nq = a.nq, nt = a.nt, out_rows = a.nq;
  size_t out0 = 0;
  if (a.counts != nullptr) {
    const int p = blockIdx.y, fq = a.pair_q[p], ft = a.pair_t[p];
    qrows = a.q + (size_t)fq * a.cap * words;
    trows = a.q + (size_t)ft * a.cap * words;
    nq = min(a.counts[fq], a.cap);
    nt = min(a.counts[ft], a.cap);
    out_rows = a.cap;
    out0 = (size_t)p * a.cap;
  }
  nt = __builtin_amdgcn_readfirstlane(nt);
  const int qi = blockIdx.x * 64 + lane;
  if (blockIdx.x * 64 >= out_rows) return;
  const int qc = qi < nq ? qi : (nq > 0 ? nq - 1 : 0);
  uint32_t qr[W ? W : 1];
  if (nq > 0) {
    if constexpr (W != 0) {
#pragma unroll
      for (int k = 0; k < W; ++k) qr[k] = qrows[(size_t)qc * W + k];
    } else {
      for (int k = 0; k < words; ++k) qs[k * 64 + lane] = qrows[(size_t)qc * words + k];
    }
  }
  uint32_t best = 0xFFFFFFFFu, second = 0xFFFFFFFFu;
  if (nq > 0) {
    for (int j = 0; j < nt; ++j) {
      const uint32_t* tp = trows + (size_t)j * words;  // wave-uniform: scalar loads
      uint32_t d = 0;
      if constexpr (W != 0) {
#pragma unroll
        for (int k = 0; k < W; ++k) d += (uint32_t)__popc(qr[k] ^ tp[k]);
      } else {
        for (int k = 0; k < words; ++k) d += (uint32_t)__popc(qs[k * 64 + lane] ^ tp[k]);
      }
      const uint32_t key = (d << 16) | (uint32_t)j;
      second = umed3(key, best, second);
      best = min(best, key);
    }
  }
  if (qi < out_rows) {
    const bool valid = qi < nq;
    const uint32_t b = valid ? best : 0xFFFFFFFFu, sc = valid ? second : 0xFFFFFFFFu;
    a.idx1[out0 + qi] = (b == 0xFFFFFFFFu) ? -1 : (int32_t)(b & 0xFFFFu);
    a.d1[out0 + qi] = (uint16_t)(b >> 16);
    a.d2[out0 + qi] = (uint16_t)(sc >> 16);
  }
}

gh_status bf_bytes_launch(gh_ctx* ctx, const BytesArgs& a, int rows, int ny) {
  const dim3 grid(gh_div_up(rows, 64), ny);
  if (a.words == 16) {
    GH_LAUNCH(ctx, "bf_match_bytes", bf_match_bytes_kernel<16>, grid, dim3(64), 0, a);
  } else {
    GH_LAUNCH(ctx, "bf_match_bytes", bf_match_bytes_kernel<0>, grid, dim3(64), (size_t)a.words * 64 * 4, a);
  }
  return GH_OK;
}
}  // namespace

// Train sets beyond 65535 rows, any descriptor width: chunks of 65532 rows folded with bf_merge_chunk_kernel exactly as the 32-byte
// path does (bf_match_any): the 16-bit train index of the search key holds one chunk, an equal distance in a later chunk never
// displaces the earlier index.  tmp: bf_chunk_tmp_bytes(nq, nt) bytes of device memory (null up to 65535 rows).
static gh_status bf_bytes_any(gh_ctx* ctx, const uint8_t* q_dev, int nq, const uint8_t* t_dev, int nt, int desc_bytes, int32_t* idx1_dev,
                              uint16_t* d1_dev, uint16_t* d2_dev, void* tmp) {
  constexpr int kChunk = 65532;
  const size_t ib = (((size_t)nq * 4) + 255) & ~(size_t)255, db = (((size_t)nq * 2) + 255) & ~(size_t)255;
  int32_t* cidx = (int32_t*)tmp;
  uint16_t* cd1 = (uint16_t*)((uint8_t*)tmp + ib);
  uint16_t* cd2 = (uint16_t*)((uint8_t*)tmp + ib + db);
  for (int base = 0; base < (nt > 0 ? nt : 1); base += kChunk) {
    const int n = nt <= 65535 ? nt : (nt - base < kChunk ? nt - base : kChunk);
    const bool first = base == 0;
    BytesArgs a{(const uint32_t*)q_dev, (const uint32_t*)(t_dev + (size_t)base * desc_bytes), nullptr, nullptr, nullptr, nq, n, 0, desc_bytes / 4,
                first ? idx1_dev : cidx, first ? d1_dev : cd1, first ? d2_dev : cd2};
    GH_TRY(bf_bytes_launch(ctx, a, nq, 1));
    if (!first)
      GH_LAUNCH(ctx, "bf_merge_chunk", bf_merge_chunk_kernel, dim3(gh_div_up(nq, 256)), dim3(256), 0, nq, idx1_dev, d1_dev, d2_dev,
                (const int32_t*)cidx, (const uint16_t*)cd1, (const uint16_t*)cd2, base);
    if (nt <= 65535) break;
  }
  return GH_OK;
}

extern "C" gh_status gh_bf_match_bytes_dev(gh_ctx* ctx, const uint8_t* q_dev, int nq, const uint8_t* t_dev, int nt, int desc_bytes,
                                           int32_t* idx1_dev, uint16_t* d1_dev, uint16_t* d2_dev) {
  if (!ctx) return GH_ERR_ARG;
  if (desc_bytes == 32) return gh_bf_match_dev(ctx, q_dev, nq, t_dev, nt, idx1_dev, d1_dev, d2_dev);
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, nq >= 0 && nt >= 0 && desc_bytes >= 8 && desc_bytes <= 256 && desc_bytes % 8 == 0);
  if (nq == 0) return GH_OK;
  GH_CHECK_ARG(ctx, q_dev && idx1_dev && d1_dev && d2_dev && (nt == 0 || t_dev));
  GH_CHECK_ARG(ctx, ((uintptr_t)q_dev & 3) == 0 && ((uintptr_t)t_dev & 3) == 0);
  void* tmp = nullptr;
  if (nt > 65535) GH_TRY(gh_scratch(ctx, bf_chunk_tmp_bytes(nq, nt), &tmp));
  return bf_bytes_any(ctx, q_dev, nq, t_dev, nt, desc_bytes, idx1_dev, d1_dev, d2_dev, tmp);
}

// host arrays in and out (what FeatureDetector::match of the plugin calls for descriptors that are not 32 bytes wide)
extern "C" gh_status gh_bf_match_bytes_host(gh_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int desc_bytes,
                                            int32_t* idx1, uint16_t* d1, uint16_t* d2) {
  if (!ctx) return GH_ERR_ARG;
  if (desc_bytes == 32) return gh_bf_match_host(ctx, q, nq, t, nt, idx1, d1, d2);
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, nq >= 0 && nt >= 0 && desc_bytes >= 8 && desc_bytes <= 256 && desc_bytes % 8 == 0);
  if (nq == 0) return GH_OK;
  GH_CHECK_ARG(ctx, q && idx1 && d1 && d2 && (nt == 0 || t));
  const size_t qb = (size_t)nq * desc_bytes, tb = (size_t)nt * desc_bytes;
  const size_t off_t = (qb + 255) & ~(size_t)255, off_i = off_t + ((tb + 255) & ~(size_t)255);
  const size_t off_d1 = off_i + (((size_t)nq * 4 + 255) & ~(size_t)255), off_d2 = off_d1 + (((size_t)nq * 2 + 255) & ~(size_t)255);
  const size_t total = off_d2 + (size_t)nq * 2;
  const size_t off_tmp = (total + 255) & ~(size_t)255;  // chunk results of a train set beyond 65535 rows (device side only)
  void *base = nullptr, *hbase = nullptr;
  GH_TRY(gh_scratch(ctx, off_tmp + bf_chunk_tmp_bytes(nq, nt), &base));
  GH_TRY(gh_pinned(ctx, total, &hbase));
  uint8_t *b = (uint8_t*)base, *hb = (uint8_t*)hbase;
  memcpy(hb, q, qb);
  if (tb) memcpy(hb + off_t, t, tb);
  GH_HIP(ctx, hipMemcpyAsync(b, hb, off_t + tb, hipMemcpyHostToDevice, ctx->stream));
  GH_TRY(bf_bytes_any(ctx, b, nq, b + off_t, nt, desc_bytes, (int32_t*)(b + off_i), (uint16_t*)(b + off_d1), (uint16_t*)(b + off_d2),
                      nt > 65535 ? b + off_tmp : nullptr));
  GH_HIP(ctx, hipMemcpyAsync(hb + off_i, b + off_i, total - off_i, hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(idx1, hb + off_i, (size_t)nq * 4);
  memcpy(d1, hb + off_d1, (size_t)nq * 2);
  memcpy(d2, hb + off_d2, (size_t)nq * 2);
  return GH_OK;
}

extern "C" gh_status gh_bf_match_pairs_bytes_dev(gh_ctx* ctx, const uint8_t* desc_dev, const int32_t* counts_dev, int cap, int desc_bytes,
                                                 const int32_t* pair_q_dev, const int32_t* pair_t_dev, int npairs,
                                                 int32_t* idx1_dev, uint16_t* d1_dev, uint16_t* d2_dev) {
  if (!ctx) return GH_ERR_ARG;
  if (desc_bytes == 32)
    return gh_bf_match_pairs_dev(ctx, desc_dev, counts_dev, cap, pair_q_dev, pair_t_dev, npairs, idx1_dev, d1_dev, d2_dev);
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, cap >= 0 && cap <= 65535 && npairs >= 0 && desc_bytes >= 8 && desc_bytes <= 256 && desc_bytes % 8 == 0);
  if (npairs == 0 || cap == 0) return GH_OK;
  GH_CHECK_ARG(ctx, desc_dev && counts_dev && pair_q_dev && pair_t_dev && idx1_dev && d1_dev && d2_dev && ((uintptr_t)desc_dev & 3) == 0);
  for (int p0 = 0; p0 < npairs; p0 += 65535) {  // grid.y is limited to 65535
    const int np = npairs - p0 < 65535 ? npairs - p0 : 65535;
    const size_t o = (size_t)p0 * cap;
    BytesArgs a{(const uint32_t*)desc_dev, nullptr, counts_dev, pair_q_dev + p0, pair_t_dev + p0, 0, 0, cap, desc_bytes / 4,
                idx1_dev + o, d1_dev + o, d2_dev + o};
    GH_TRY(bf_bytes_launch(ctx, a, cap, np));
  }
  return GH_OK;
}

extern "C" gh_status gh_bf_match_band_pairs_dev(gh_ctx* ctx, const uint8_t* desc_dev, const gh_keypoint* kps_dev,
                                                const int32_t* counts_dev, int cap, const int32_t* pair_q_dev,
                                                const int32_t* pair_t_dev, int npairs, float band_per_size,
                                                int32_t* idx1_dev, uint16_t* d1_dev, uint16_t* d2_dev) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, cap >= 0 && cap <= 65535 && npairs >= 0 && band_per_size >= 0.f);
  if (npairs == 0 || cap == 0) return GH_OK;
  GH_CHECK_ARG(ctx, desc_dev && kps_dev && counts_dev && pair_q_dev && pair_t_dev && idx1_dev && d1_dev && d2_dev);
  for (int p0 = 0; p0 < npairs; p0 += 65535) {
    const int np = npairs - p0 < 65535 ? npairs - p0 : 65535;
    const size_t o = (size_t)p0 * cap;
    GH_LAUNCH(ctx, "bf_match_band", bf_match_band_pairs_kernel, dim3(gh_div_up(cap, 64), np), dim3(64), 0,
              (const uint32_t*)desc_dev, kps_dev, counts_dev, cap, pair_q_dev + p0, pair_t_dev + p0, band_per_size,
              idx1_dev + o, d1_dev + o, d2_dev + o);
  }
  return GH_OK;
}

extern "C" gh_status gh_match_mask_dev(gh_ctx* ctx, const int32_t* idx1_dev, const uint16_t* d1_dev,
                                       const uint16_t* d2_dev, int nq, const int32_t* back_idx1_dev, int nt,
                                       int max_dist, int ratio_num, int ratio_den, int cross_check,
                                       uint8_t* keep_dev) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, nq >= 0 && nt >= 0);
  if (nq == 0) return GH_OK;
  GH_CHECK_ARG(ctx, idx1_dev && d1_dev && d2_dev && keep_dev && (!cross_check || back_idx1_dev));
  GH_LAUNCH(ctx, "match_mask", match_mask_kernel, dim3(gh_div_up(nq, 256)), dim3(256), 0, idx1_dev, d1_dev, d2_dev,
            nq, back_idx1_dev, nt, max_dist, ratio_num, ratio_den, cross_check, keep_dev);
  return GH_OK;
}

extern "C" gh_status gh_bf_valu_probe(gh_ctx* ctx, double* pairs_per_s) {
  if (!ctx || !pairs_per_s) return GH_ERR_ARG;
  GH_ENTER(ctx);
  void* out = nullptr;
  GH_TRY(gh_scratch(ctx, 256, &out));
  const int iters = 4096, blocks = 256 * 16;
  hipEvent_t e0, e1;
  GH_HIP(ctx, hipEventCreate(&e0));
  GH_HIP(ctx, hipEventCreate(&e1));
  hipLaunchKernelGGL(valu_probe_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (uint32_t*)out, 16, 12345u);
  GH_HIP(ctx, hipEventRecord(e0, ctx->stream));
  hipLaunchKernelGGL(valu_probe_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (uint32_t*)out, iters, 12345u);
  GH_HIP(ctx, hipEventRecord(e1, ctx->stream));
  GH_HIP(ctx, hipEventSynchronize(e1));
  float ms = 0.f;
  GH_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  double pairs = (double)blocks * 256.0 * iters * 4.0;
  *pairs_per_s = pairs / (ms * 1e-3);
  return GH_OK;
}
