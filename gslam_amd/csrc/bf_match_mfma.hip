// Brute-force 256-bit Hamming matcher, exact integer MFMA formulation (an experiment reported BESIDE
// the popcount kernel of bf_match.hip, which stays the contract path of north_star -- "popcount, not a dense contraction").
//
// Same semantics, bit for bit (GSLAM/core/Vocabulary.h:485-491 distance, :1712-1725 first strict minimum):
//   hamming(a, b) - |a| = #(b = 1, a = 0) - #(b = 1, a = 1) = sum over the 256 bits of  A_k * B_k / 4096
// with the query expanded to A_k = -64 (bit set) / +64 (bit clear) and the train row to B_k = 64 / 0, one int8 per bit.
// v_mfma_i32_16x16x64_i8 accumulates that sum exactly (|sum| <= 2^20), four MFMAs per 16 x 16 tile of pairs, and the
// accumulator STARTS at  C = (256 << 12) | t  (t = number of the 16-row train tile), so that what the matrix core
// returns already is the search key
//     key = 4096 * (hamming - |a_i| + 256) + t
// -- for a fixed query and a fixed lane (train rows j = 16 t + c of one column c) the order of the keys is the order of
// (hamming, j).  The VALU is left with two instructions per pair, v_med3_u32 + v_min_u32 (best / second best), against
// nineteen in the popcount kernel; |a_i| comes back in when the winner is written, and (key << 4) | c is the
// (distance << 16 | j) key of the popcount kernel, which is what the 16 columns are merged on.  t < 4096 covers the
// contract's cap <= 65535.
//
// Layout.  A workgroup is 8 waves; a wave owns 64 queries (4 tiles of 16) for the whole train set.  Lane l = 16 g + c:
// for the A operand c is the query row of the tile, for B the train row; g selects 8 of the 32 descriptor bytes (bytes
// 8 g .. 8 g + 7).  The k dimension of the four MFMAs of a tile is laid out as [chunk m][lane group g][16 slots] = the
// bits of those 8 bytes -- ANY assignment of descriptor bits to k slots is correct as long as A and B use the same one.
// D layout of the instruction: lane (g, c) holds rows 4 g + r (r = 0..3), column c: best / second-best are tracked per
// lane over the train rows = c (mod 16) and merged over the 16 lanes of a DPP row at the end.
//
// The train side is expanded ONCE per workgroup: in every round of 8 train tiles wave w turns tile w of the NEXT round
// into operand bytes (8 look-ups of a 256-entry byte -> 8-byte table in LDS per lane) and parks the four 16-byte chunks in
// an LDS stage in exactly the lane order the MFMA wants, so every wave fetches a tile's B operand with four conflict-free
// ds_read_b128 and no address arithmetic (two stages of 8 x 4 KB, one workgroup barrier per round).  What the VALU does
// per MFMA is then the two fold instructions and little else -- measured on gfx950 (tools/mfma_probe.hip): an MFMA of
// this shape occupies the matrix core for 16 clocks and the issue port for 8, every VALU instruction beside it costs 4.
#include "common.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kTileBytes = 4096;              // one expanded train tile: 4 chunks x 64 lanes x 16 bytes
constexpr uint32_t kInvalidKey = 0x0FF00000u;  // above every real key (< 2^22) even after the MFMA subtracts up to 2^20

// median of three as plain min / max so that the compiler selects v_med3_u32 ITSELF: an inline-asm consumer of an MFMA
// result is invisible to the hazard recogniser (no wait states get inserted for it -- measured: stale second-best keys)
__device__ __forceinline__ uint32_t umed3m(uint32_t a, uint32_t b, uint32_t c) {
  return max(min(a, b), min(max(a, b), c));
}

// byte `i` (0..7) of the lane's 8 descriptor bytes -> 8 operand bytes of 0 / 0x40 through the table
__device__ __forceinline__ uint2 lut_byte(const uint8_t* __restrict__ lut, uint2 w, int i) {
  const uint32_t word = i < 4 ? w.x : w.y;
  const uint32_t sh = (i & 3) == 0 ? word << 3 : (i & 3) == 1 ? word >> 5 : (i & 3) == 2 ? word >> 13 : word >> 21;
  return *reinterpret_cast<const uint2*>(lut + (sh & 0x7F8u));
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

// kQT query tiles of 16 per wave, kWaves waves per workgroup = train tiles per round
template <int kQT, int kWaves>
__global__ __launch_bounds__(64 * kWaves) void bf_match_pairs_mfma_kernel(
    const uint8_t* __restrict__ desc, const int32_t* __restrict__ counts, int cap, const int32_t* __restrict__ pair_q,
    const int32_t* __restrict__ pair_t, int npairs, int32_t* __restrict__ idx1, uint16_t* __restrict__ d1,
    uint16_t* __restrict__ d2) {
  constexpr int kWaveQ = 16 * kQT, kStageBytes = kWaves * kTileBytes, kLutOffset = 2 * kStageBytes;
  constexpr int kSlots = 4 * kQT;  // MFMAs per tile
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint8_t* const lut = lds + kLutOffset;
  if (threadIdx.x < 256) {
    // byte v -> 8 bytes, byte k = 0x40 * bit k of v
    const uint32_t v = threadIdx.x;
    const uint32_t lo = (((v & 0xFu) * 0x00204081u) & 0x01010101u) << 6, hi = ((((v >> 4) & 0xFu) * 0x00204081u) & 0x01010101u) << 6;
    *reinterpret_cast<uint2*>(lut + 8 * v) = make_uint2(lo, hi);
  }
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int q0 = (blockIdx.x * kWaves + wv) * kWaveQ;
  const uint32_t lane16 = 16u * (uint32_t)lane;
  // a workgroup keeps its table and walks over frame pairs blockIdx.y, blockIdx.y + gridDim.y, ...
  for (int p = blockIdx.y; p < npairs; p += gridDim.y) {
    const int fq = pair_q[p], ft = pair_t[p];
    int nq = counts[fq], nt = counts[ft];
    nq = nq < cap ? nq : cap;
    nt = nt < cap ? nt : cap;
    const size_t out_base = (size_t)p * cap;
    const bool active = q0 < nq;  // wave-uniform: this wave has at least one valid query
    if (!active || nt == 0) {     // nothing to match: the defined empty result for this wave's rows
      for (int i = q0 + lane; i < min(q0 + kWaveQ, cap); i += 64) {
        idx1[out_base + i] = -1;
        d1[out_base + i] = 65535;
        d2[out_base + i] = 65535;
      }
    }
    if (nt == 0 || blockIdx.x * kWaves * kWaveQ >= nq) continue;  // workgroup-uniform: no barrier is skipped by a part of it
    const uint8_t* qd = desc + (size_t)fq * cap * 32;
    const uint8_t* td = desc + (size_t)ft * cap * 32;
    const int n_tiles = (nt + 15) >> 4, n_rounds = (n_tiles + kWaves - 1) / kWaves;
    auto load_rows = [&](int t) {
      const uint32_t off = min((uint32_t)(16 * t + c), (uint32_t)(cap - 1)) * 32u + 8u * (uint32_t)g;  // (cap * 32 < 2^21)
      return *reinterpret_cast<const uint2*>(td + off);
    };
    // this wave's tile of round `r` -> stage (r & 1), slot wv
    auto produce = [&](int r, uint2 w) {
      uint8_t* dst = lds + (r & 1) * kStageBytes + wv * kTileBytes + lane16;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const uint2 e0 = lut_byte(lut, w, 2 * m), e1 = lut_byte(lut, w, 2 * m + 1);
        *reinterpret_cast<v4i*>(dst + 1024 * m) = v4i{(int)e0.x, (int)e0.y, (int)e1.x, (int)e1.y};
      }
    };
    __syncthreads();  // the table is there; nobody still reads a stage of the previous pair
    uint2 wnext = load_rows(kWaves + wv);  // my tile of round 1, requested a round ahead
    produce(0, load_rows(wv));
    if (!active) {
      // a wave without queries still owes its tiles and its barriers
      __syncthreads();
      for (int r = 0; r < n_rounds; ++r) {
        const uint2 w = wnext;
        wnext = load_rows(kWaves * (r + 2) + wv);
        produce(r + 1, w);
        __syncthreads();
      }
      continue;
    }
    // ---- queries: 8 bytes per lane per tile, expanded once to -64 (bit set) / +64
    v4i a[kQT][4];
#pragma unroll
    for (int qt = 0; qt < kQT; ++qt) {
      const int row = min(q0 + 16 * qt + c, cap - 1);
      const uint2 w = *reinterpret_cast<const uint2*>(qd + (size_t)row * 32 + 8 * g);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const uint2 e0 = lut_byte(lut, w, 2 * m), e1 = lut_byte(lut, w, 2 * m + 1);
        a[qt][m] = v4i{(int)((e0.x << 1) | 0x40404040u), (int)((e0.y << 1) | 0x40404040u), (int)((e1.x << 1) | 0x40404040u),
                       (int)((e1.y << 1) | 0x40404040u)};  // 0x40 -> 0xC0 (-64), 0 -> 0x40 (+64)
      }
    }
    uint32_t b1[kQT][4], b2[kQT][4];
#pragma unroll
    for (int qt = 0; qt < kQT; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) b1[qt][r] = b2[qt][r] = 0xFFFFFFFFu;
    // The loop is software-pipelined ACROSS tiles and the issue order is pinned with sched_barrier: slot s of a tile is
    // one MFMA of THIS tile (chunk-major, so consecutive MFMAs never depend on each other) followed by the two VALU
    // instructions that fold one accumulator register of the PREVIOUS tile into best / second best.  The B operand is
    // double-buffered in registers: chunk m of the next tile is fetched from the stage in the slot of the last MFMA that
    // reads chunk m of this one (a single register set measured 2 % slower).  The other slots carry the wave's share of
    // the next round (8 look-ups, 4 stores).
    v4i bx[4], by[4], acc0[kQT], acc1[kQT];
#pragma unroll
    for (int qt = 0; qt < kQT; ++qt) acc1[qt] = v4i{-1, -1, -1, -1};  // "previous tile" of tile 0: keys that change nothing
    auto fold = [&](const v4i(&acc)[kQT], int k) {
      const int qt = k >> 2, r = k & 3;
      const uint32_t key = (uint32_t)acc[qt][r];
      b2[qt][r] = umed3m(key, b1[qt][r], b2[qt][r]);
      // med3 first and the min not shared with the one inside the med3 pattern (the empty asm hides the equality): b1 is
      // then updated in place; otherwise every tile starts with 4 kQT register copies of the loop-carried b1
      uint32_t old1 = b1[qt][r];
      asm("" : "+v"(old1));
      __builtin_amdgcn_sched_barrier(0);
      b1[qt][r] = min(old1, key);
    };
    __syncthreads();  // stage 0 holds round 0
#pragma unroll
    for (int m = 0; m < 4; ++m) bx[m] = *reinterpret_cast<const v4i*>(lds + lane16 + 1024 * m);
    for (int r = 0; r < n_rounds; ++r) {
      const uint8_t* const rd = lds + (r & 1) * kStageBytes + lane16;          // this round's stage, as this lane reads it
      const uint8_t* const rd_next = lds + ((r + 1) & 1) * kStageBytes + lane16;
      uint8_t* const wr_next = lds + ((r + 1) & 1) * kStageBytes + wv * kTileBytes + lane16;
      const uint2 wprod = wnext;  // my tile of round r + 1
      wnext = load_rows(kWaves * (r + 2) + wv);
      v4i e[2];
#pragma unroll
      for (int tau = 0; tau < kWaves; ++tau) {
        v4i(&cur)[4] = (tau & 1) ? by : bx;
        v4i(&nxt)[4] = (tau & 1) ? bx : by;
        v4i(&acc)[kQT] = (tau & 1) ? acc1 : acc0;
        const v4i(&prev)[kQT] = (tau & 1) ? acc0 : acc1;
        const int t = kWaves * r + tau;
        // train rows past the count start from a key above every real one
        const uint32_t kb = (16 * t + c < nt ? (256u << 12) : kInvalidKey) + (uint32_t)t;
        const v4i cinit = {(int)kb, (int)kb, (int)kb, (int)kb};
        if (tau == kWaves - 1) __syncthreads();  // every wave has parked its tile of round r + 1; stage (r & 1) is free after this tile
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int qt = 0; qt < kQT; ++qt) {
            const int s = m * kQT + qt;
            acc[qt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[qt][m], cur[m], m == 0 ? cinit : acc[qt], 0, 0, 0);
            fold(prev, s);
            if (qt == kQT - 1) {
              // chunk m is through for this tile: fetch the next tile's (the first tile of the next round sits in the other stage)
              nxt[m] = *reinterpret_cast<const v4i*>(tau + 1 < kWaves ? rd + kTileBytes * (tau + 1) + 1024 * m : rd_next + 1024 * m);
            } else {
              // my share of the next round, one item per free slot: 4 look-ups, 2 stores, 4 look-ups, 2 stores
              const int item = tau * (kSlots - 4) + m * (kQT - 1) + qt;
              if (item < 12) {
                const int half = item / 6, k = item % 6;  // half 0: descriptor bytes 0..3 -> chunks 0, 1; half 1: bytes 4..7 -> chunks 2, 3
                if (k < 4) {
                  const uint2 h = lut_byte(lut, wprod, 4 * half + k);
                  e[k >> 1][2 * (k & 1)] = (int)h.x;
                  e[k >> 1][2 * (k & 1) + 1] = (int)h.y;
                } else {
                  *reinterpret_cast<v4i*>(wr_next + 1024 * (2 * half + k - 4)) = e[k - 4];
                }
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    // the last tile's accumulators are still unfolded
#pragma unroll
    for (int k = 0; k < 4 * kQT; ++k) fold(acc1, k);
    // ---- (key << 4) | c = (distance' << 16) | j; merge over the 16 lanes of each DPP row (the columns), then lane c == 0
    // of row g writes query rows 4 g + r
#pragma unroll
    for (int qt = 0; qt < kQT; ++qt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        b1[qt][r] = (b1[qt][r] << 4) | (uint32_t)c;
        b2[qt][r] = (b2[qt][r] << 4) | (uint32_t)c;  // 0xFFFFFFFF stays above every real key
        merge_dpp<0x128>(b1[qt][r], b2[qt][r]);  // row_ror:8
        merge_dpp<0x124>(b1[qt][r], b2[qt][r]);  // row_ror:4
        merge_dpp<0x122>(b1[qt][r], b2[qt][r]);  // row_ror:2
        merge_dpp<0x121>(b1[qt][r], b2[qt][r]);  // row_ror:1
      }
      // |a| of the query rows comes back in here (read again rather than carried through the loop in registers)
      const int row = min(q0 + 16 * qt + c, cap - 1);
      const uint2 w = *reinterpret_cast<const uint2*>(qd + (size_t)row * 32 + 8 * g);
      int pa = __popc(w.x) + __popc(w.y);
      pa += __shfl_xor(pa, 16);
      pa += __shfl_xor(pa, 32);  // every lane (*, c) holds |a| of row c of tile qt
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int pq = __shfl(pa, 4 * g + r);
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
  }  // pairs
}

template <int kQT, int kWaves>
gh_status launch_variant(gh_ctx* ctx, const uint8_t* desc_dev, const int32_t* counts_dev, int cap, const int32_t* pair_q_dev,
                         const int32_t* pair_t_dev, int npairs, int32_t* idx1_dev, uint16_t* d1_dev, uint16_t* d2_dev) {
  constexpr int kLdsBytes = 2 * kWaves * kTileBytes + 2048;
  static bool lds_attr_set[64] = {};  // per device: the attribute belongs to the code object loaded there
  const int dev = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
  if (!lds_attr_set[dev]) {
    GH_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(bf_match_pairs_mfma_kernel<kQT, kWaves>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    lds_attr_set[dev] = true;
  }
  // pairs are walked by persistent workgroups (one per CU fits); four times as many workgroups as CUs so that the
  // hardware's dynamic dispatch evens out ragged pairs
  const int gx = gh_div_up(cap, 16 * kQT * kWaves);
  int gy = (4 * (ctx->cu_count > 0 ? ctx->cu_count : 256) + gx - 1) / gx;
  gy = gy < npairs ? gy : npairs;
  GH_LAUNCH(ctx, "bf_match_pairs_mfma", (bf_match_pairs_mfma_kernel<kQT, kWaves>), dim3(gx, gy), dim3(64 * kWaves), kLdsBytes,
            desc_dev, counts_dev, cap, pair_q_dev, pair_t_dev, npairs, idx1_dev, d1_dev, d2_dev);
  return GH_OK;
}

}  // namespace

// Same contract as gh_bf_match_pairs_dev (include/gslam_hip.h), different arithmetic route.
extern "C" gh_status gh_bf_match_pairs_mfma_dev(gh_ctx* ctx, const uint8_t* desc_dev, const int32_t* counts_dev, int cap,
                                                const int32_t* pair_q_dev, const int32_t* pair_t_dev, int npairs,
                                                int32_t* idx1_dev, uint16_t* d1_dev, uint16_t* d2_dev) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, cap >= 0 && cap <= 65535 && npairs >= 0);
  if (npairs == 0 || cap == 0) return GH_OK;
  GH_CHECK_ARG(ctx, desc_dev && counts_dev && pair_q_dev && pair_t_dev && idx1_dev && d1_dev && d2_dev);
  GH_CHECK_ARG(ctx, ((uintptr_t)desc_dev & 7) == 0);
  // 4 query tiles x 8 waves: two waves per SIMD.  2 x 16 (four waves per SIMD, 128 registers) measured 9 % slower: the
  // per-tile work is amortised over half as many MFMAs
  return launch_variant<4, 8>(ctx, desc_dev, counts_dev, cap, pair_q_dev, pair_t_dev, npairs, idx1_dev, d1_dev, d2_dev);
}
