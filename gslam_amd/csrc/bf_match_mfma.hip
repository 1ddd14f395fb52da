// Brute-force 256-bit Hamming matcher, exact integer MFMA formulation (an experiment reported BESIDE
// the popcount kernel of bf_match.hip, which stays the contract path of north_star -- "popcount, not a dense contraction").
//
// Same semantics, bit for bit (GSLAM/core/Vocabulary.h:485-491 distance, :1712-1725 first strict minimum):
//   hamming(a, b) = |a| + |b| - 2 |a & b|, and |a & b| is a dot product of the two descriptors expanded to one byte per bit.
// Queries are expanded to {0, -1}, train rows to {0, 1}: v_mfma_i32_16x16x64_i8 accumulates dot = -|a & b| exactly
// (|dot| <= 256), four MFMAs per 16 x 16 tile of pairs.  Per pair the VALU then does three instructions instead of
// nineteen:  key = (dot << 17) + keybase_j  with keybase_j = ((|b_j| + 256) << 16) | j   (v_lshl_add_u32), i.e.
// key = ((hamming - |a_i| + 256) << 16) | j -- for a fixed query the order of the keys is the order of (hamming, j) --
// then v_med3_u32 + v_min_u32 as in the popcount kernel.  |a_i| comes back in when the winner is written.
//
// Layout.  A wave owns 16 * kQT queries for the whole train set.  Lane l = 16 g + c: for the A operand c is the query row
// of the tile, for B the train row; g selects 8 of the 32 descriptor bytes (bytes 8 g .. 8 g + 7).  The k dimension of the
// four MFMAs of a tile is laid out as [chunk m][lane group g][16 slots] = bit 16 m' ... of those 8 bytes -- ANY assignment
// of descriptor bits to k slots is correct as long as A and B use the same one, so each lane expands exactly the 8 bytes it
// loaded (one global_load_dwordx2 per train row per lane, 512 contiguous bytes per tile).  Expansion of a byte into 8
// operand bytes is one 8-byte LDS read from a 256-entry table (the VALU only forms the address); the query side is
// expanded once per wave and multiplied by 0xFF.  D layout of the instruction: lane (g, c) holds rows 4 g + r (r = 0..3),
// column c: best / second-best are tracked per lane over the train rows = c (mod 16) and merged over the 16 lanes of a
// DPP row at the end.
#include "common.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kQT = 4;                 // query tiles of 16 per wave
constexpr int kWaveQ = 16 * kQT;       // 64 queries per wave
constexpr int kWavesPerWg = 4;

__device__ __forceinline__ uint32_t umed3m(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// 8 descriptor bytes (two dwords) -> four 16-byte operand chunks through the LDS table (byte -> 8 bytes of 0 / 1)
__device__ __forceinline__ void expand8(const uint2* __restrict__ lut, uint32_t lo, uint32_t hi, v4i (&out)[4]) {
  // chunk m takes descriptor bytes 2 m and 2 m + 1 of the lane's 8
  const uint2 e0 = lut[lo & 0xFFu], e1 = lut[(lo >> 8) & 0xFFu], e2 = lut[(lo >> 16) & 0xFFu], e3 = lut[lo >> 24];
  const uint2 e4 = lut[hi & 0xFFu], e5 = lut[(hi >> 8) & 0xFFu], e6 = lut[(hi >> 16) & 0xFFu], e7 = lut[hi >> 24];
  out[0] = v4i{(int)e0.x, (int)e0.y, (int)e1.x, (int)e1.y};
  out[1] = v4i{(int)e2.x, (int)e2.y, (int)e3.x, (int)e3.y};
  out[2] = v4i{(int)e4.x, (int)e4.y, (int)e5.x, (int)e5.y};
  out[3] = v4i{(int)e6.x, (int)e6.y, (int)e7.x, (int)e7.y};
}

// merge (b1, b2) with the pair of the lane `other` positions away inside the 16-lane DPP row
template <int CTRL>
__device__ __forceinline__ void merge_dpp(uint32_t& b1, uint32_t& b2) {
  const uint32_t o1 = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)b1, CTRL, 0xf, 0xf, false);
  const uint32_t o2 = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)b2, CTRL, 0xf, 0xf, false);
  const uint32_t hi = max(b1, o1);
  b1 = min(b1, o1);
  b2 = min(hi, min(b2, o2));
}

__global__ __launch_bounds__(64 * kWavesPerWg) void bf_match_pairs_mfma_kernel(
    const uint8_t* __restrict__ desc, const int32_t* __restrict__ counts, int cap, const int32_t* __restrict__ pair_q,
    const int32_t* __restrict__ pair_t, int32_t* __restrict__ idx1, uint16_t* __restrict__ d1, uint16_t* __restrict__ d2) {
  __shared__ uint2 lut[256];
  {
    // byte v -> 8 bytes, byte k = bit k of v
    const uint32_t v = threadIdx.x;
    const uint32_t lo = ((v & 0xFu) * 0x00204081u) & 0x01010101u, hi = (((v >> 4) & 0xFu) * 0x00204081u) & 0x01010101u;
    lut[v] = make_uint2(lo, hi);
  }
  __syncthreads();
  const int p = blockIdx.y;
  const int fq = pair_q[p], ft = pair_t[p];
  int nq = counts[fq], nt = counts[ft];
  nq = nq < cap ? nq : cap;
  nt = nt < cap ? nt : cap;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int q0 = (blockIdx.x * kWavesPerWg + wv) * kWaveQ;
  const size_t out_base = (size_t)p * cap;
  if (q0 >= cap) return;
  if (nt == 0 || q0 >= nq) {  // nothing to match against, or no valid query in this wave: the defined empty result
    for (int i = q0 + lane; i < min(q0 + kWaveQ, cap); i += 64) {
      idx1[out_base + i] = -1;
      d1[out_base + i] = 65535;
      d2[out_base + i] = 65535;
    }
    return;
  }
  const uint8_t* qd = desc + (size_t)fq * cap * 32;
  const uint8_t* td = desc + (size_t)ft * cap * 32;
  // ---- queries: 8 bytes per lane per tile, expanded once, as {0, -1}; |a| of row c of every tile
  v4i a[kQT][4];
  int pa[kQT];
#pragma unroll
  for (int qt = 0; qt < kQT; ++qt) {
    const int row = min(q0 + 16 * qt + c, cap - 1);
    const uint2 w = *reinterpret_cast<const uint2*>(qd + (size_t)row * 32 + 8 * g);
    expand8(lut, w.x, w.y, a[qt]);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) a[qt][m][e] = (int)((uint32_t)a[qt][m][e] * 0xFFu);  // 0x01 -> 0xFF per byte, no carries
    int pc = __popc(w.x) + __popc(w.y);
    pc += __shfl_xor(pc, 16);
    pc += __shfl_xor(pc, 32);
    pa[qt] = pc;  // every lane (c, *) holds |a| of row c of tile qt
  }
  uint32_t b1[kQT][4], b2[kQT][4];
#pragma unroll
  for (int qt = 0; qt < kQT; ++qt)
#pragma unroll
    for (int r = 0; r < 4; ++r) b1[qt][r] = b2[qt][r] = 0xFFFFFFFFu;

  const int n_tiles = (nt + 15) >> 4;
  // software pipeline: the train rows of tile t + 1 are on their way while tile t is multiplied
  uint2 nxt = make_uint2(0u, 0u);
  {
    const int row = min(c, cap - 1);
    nxt = *reinterpret_cast<const uint2*>(td + (size_t)row * 32 + 8 * g);
  }
  for (int t = 0; t < n_tiles; ++t) {
    const uint2 w = nxt;
    if (t + 1 < n_tiles) {
      const int row = min(16 * (t + 1) + c, cap - 1);
      nxt = *reinterpret_cast<const uint2*>(td + (size_t)row * 32 + 8 * g);
    }
    v4i b[4];
    expand8(lut, w.x, w.y, b);
    int pb = __popc(w.x) + __popc(w.y);
    pb += __shfl_xor(pb, 16);
    pb += __shfl_xor(pb, 32);
    const int j = 16 * t + c;
    // train rows past the count: a key above every real one (real keys stay below 0x0300 << 16)
    const uint32_t keybase = j < nt ? (((uint32_t)(pb + 256) << 16) | (uint32_t)j) : (0xFFFF0000u | (uint32_t)(j & 0xFFFF));
#pragma unroll
    for (int qt = 0; qt < kQT; ++qt) {
      v4i acc = {0, 0, 0, 0};
#pragma unroll
      for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[qt][m], b[m], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        uint32_t key;
        asm("v_lshl_add_u32 %0, %1, 17, %2" : "=v"(key) : "v"(acc[r]), "v"(keybase));
        b2[qt][r] = umed3m(key, b1[qt][r], b2[qt][r]);
        b1[qt][r] = min(b1[qt][r], key);
      }
    }
  }
  // ---- merge over the 16 lanes of each DPP row (the columns), then lane c == 0 of row g writes query rows 4 g + r
#pragma unroll
  for (int qt = 0; qt < kQT; ++qt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      merge_dpp<0x128>(b1[qt][r], b2[qt][r]);  // row_ror:8
      merge_dpp<0x124>(b1[qt][r], b2[qt][r]);  // row_ror:4
      merge_dpp<0x122>(b1[qt][r], b2[qt][r]);  // row_ror:2
      merge_dpp<0x121>(b1[qt][r], b2[qt][r]);  // row_ror:1
    }
    // |a| of query row 4 g + r lives in the lanes with c = 4 g + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int pq = __shfl(pa[qt], 4 * g + r);
      const int qi = q0 + 16 * qt + 4 * g + r;
      if (c == 0 && qi < cap) {
        const uint32_t k1 = b1[qt][r], k2 = b2[qt][r];
        const bool valid = qi < nq;
        const bool has2 = (k2 >> 16) < 0x8000u;
        idx1[out_base + qi] = valid ? (int32_t)(k1 & 0xFFFFu) : -1;
        d1[out_base + qi] = valid ? (uint16_t)((int)(k1 >> 16) + pq - 256) : (uint16_t)65535;
        d2[out_base + qi] = valid && has2 ? (uint16_t)((int)(k2 >> 16) + pq - 256) : (uint16_t)65535;
      }
    }
  }
}

}  // namespace

// Same contract as gh_bf_match_pairs_dev (include/gslam_hip.h), different arithmetic route; cap must be a multiple of 1.
extern "C" gh_status gh_bf_match_pairs_mfma_dev(gh_ctx* ctx, const uint8_t* desc_dev, const int32_t* counts_dev, int cap,
                                                const int32_t* pair_q_dev, const int32_t* pair_t_dev, int npairs,
                                                int32_t* idx1_dev, uint16_t* d1_dev, uint16_t* d2_dev) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, cap >= 0 && cap <= 65535 && npairs >= 0);
  if (npairs == 0 || cap == 0) return GH_OK;
  GH_CHECK_ARG(ctx, desc_dev && counts_dev && pair_q_dev && pair_t_dev && idx1_dev && d1_dev && d2_dev);
  GH_CHECK_ARG(ctx, ((uintptr_t)desc_dev & 7) == 0 && npairs <= 65535 * 16);
  // pairs ride on grid.y (65535 at most per launch)
  for (int p0 = 0; p0 < npairs; p0 += 65535) {
    const int np = npairs - p0 < 65535 ? npairs - p0 : 65535;
    GH_LAUNCH(ctx, "bf_match_pairs_mfma", bf_match_pairs_mfma_kernel, dim3(gh_div_up(cap, kWaveQ * kWavesPerWg), np),
              dim3(64 * kWavesPerWg), 0, desc_dev, counts_dev, cap, pair_q_dev + p0, pair_t_dev + p0, idx1_dev + (size_t)p0 * cap,
              d1_dev + (size_t)p0 * cap, d2_dev + (size_t)p0 * cap);
  }
  return GH_OK;
}
