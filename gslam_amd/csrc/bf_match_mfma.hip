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
// Layout.  A wave owns 16 * QT queries for the whole train set.  Lane l = 16 g + c: for the A operand c is the query row
// of the tile, for B the train row; g selects 8 of the 32 descriptor bytes (bytes 8 g .. 8 g + 7).  The k dimension of the
// four MFMAs of a tile is laid out as [chunk m][lane group g][16 slots] = the bits of those 8 bytes -- ANY assignment
// of descriptor bits to k slots is correct as long as A and B use the same one, so each lane expands exactly the 8 bytes it
// loaded (one global_load_dwordx2 per train row per lane, 512 contiguous bytes per tile).  Expansion of a byte into 8
// operand bytes is one 8-byte LDS read from a 256-entry table that is REPLICATED 32 times, entry (v, r) at
// byte 256 v + 8 r with r = lane & 31: every lane of a ds_read_b64 lane group then owns its own pair of banks and the read
// is conflict-free whatever the descriptor bytes are (64 KB of LDS, one workgroup of 8 waves per CU).  D layout of the
// instruction: lane (g, c) holds rows 4 g + r (r = 0..3), column c: best / second-best are tracked per lane over the train
// rows = c (mod 16) and merged over the 16 lanes of a DPP row at the end.
#include "common.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kReplicas = 32;
constexpr uint32_t kInvalidKey = 0x0FF00000u;  // above every real key (< 2^22) even after the MFMA subtracts up to 2^20

// median of three as plain min / max so that the compiler selects v_med3_u32 ITSELF: an inline-asm consumer of an MFMA
// result is invisible to the hazard recogniser (no wait states get inserted for it -- measured: stale second-best keys)
__device__ __forceinline__ uint32_t umed3m(uint32_t a, uint32_t b, uint32_t c) {
  return max(min(a, b), min(max(a, b), c));
}

// 8 descriptor bytes (two dwords) -> four 16-byte operand chunks through the LDS table (byte -> 8 bytes of 0 / 0x40);
// `lane_off` = 8 * (lane & 31), the lane's replica
__device__ __forceinline__ void expand8(const uint8_t* __restrict__ lut, uint32_t lane_off, uint32_t lo, uint32_t hi,
                                        v4i (&out)[4]) {
  // chunk m takes descriptor bytes 2 m and 2 m + 1 of the lane's 8
  auto rd = [&](uint32_t shifted) { return *reinterpret_cast<const uint2*>(lut + ((shifted & 0xFF00u) | lane_off)); };
  const uint2 e0 = rd(lo << 8), e1 = rd(lo), e2 = rd(lo >> 8), e3 = rd(lo >> 16);
  const uint2 e4 = rd(hi << 8), e5 = rd(hi), e6 = rd(hi >> 8), e7 = rd(hi >> 16);
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

template <int QT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void bf_match_pairs_mfma_kernel(
    const uint8_t* __restrict__ desc, const int32_t* __restrict__ counts, int cap, const int32_t* __restrict__ pair_q,
    const int32_t* __restrict__ pair_t, int npairs, int32_t* __restrict__ idx1, uint16_t* __restrict__ d1,
    uint16_t* __restrict__ d2) {
  constexpr int kWaveQ = 16 * QT;
  __shared__ __attribute__((aligned(16))) uint8_t lut[256 * kReplicas * 8];
  for (uint32_t i = threadIdx.x; i < 256u * kReplicas; i += 64 * WAVES) {
    // byte v -> 8 bytes, byte k = 0x40 * bit k of v; 32 copies side by side
    const uint32_t v = i >> 5;
    const uint32_t lo = (((v & 0xFu) * 0x00204081u) & 0x01010101u) << 6, hi = ((((v >> 4) & 0xFu) * 0x00204081u) & 0x01010101u) << 6;
    *reinterpret_cast<uint2*>(lut + 8 * i) = make_uint2(lo, hi);
  }
  __syncthreads();
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const uint32_t lane_off = 8u * (lane & 31);
  const int q0 = (blockIdx.x * WAVES + wv) * kWaveQ;
  if (q0 >= cap) return;
  // a workgroup keeps its table and walks over frame pairs blockIdx.y, blockIdx.y + gridDim.y, ...
  for (int p = blockIdx.y; p < npairs; p += gridDim.y) {
  const int fq = pair_q[p], ft = pair_t[p];
  int nq = counts[fq], nt = counts[ft];
  nq = nq < cap ? nq : cap;
  nt = nt < cap ? nt : cap;
  const size_t out_base = (size_t)p * cap;
  if (nt == 0 || q0 >= nq) {  // nothing to match against, or no valid query in this wave: the defined empty result
    for (int i = q0 + lane; i < min(q0 + kWaveQ, cap); i += 64) {
      idx1[out_base + i] = -1;
      d1[out_base + i] = 65535;
      d2[out_base + i] = 65535;
    }
    continue;
  }
  const uint8_t* qd = desc + (size_t)fq * cap * 32;
  const uint8_t* td = desc + (size_t)ft * cap * 32;
  // ---- queries: 8 bytes per lane per tile, expanded once to -64 (bit set) / +64; |a| of row c of every tile
  v4i a[QT][4];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int row = min(q0 + 16 * qt + c, cap - 1);
    const uint2 w = *reinterpret_cast<const uint2*>(qd + (size_t)row * 32 + 8 * g);
    expand8(lut, lane_off, w.x, w.y, a[qt]);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) a[qt][m][e] = (int)(((uint32_t)a[qt][m][e] << 1) | 0x40404040u);  // 0x40 -> 0xC0, 0 -> 0x40
  }
  uint32_t b1[QT][4], b2[QT][4];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int r = 0; r < 4; ++r) b1[qt][r] = b2[qt][r] = 0xFFFFFFFFu;

  const int n_tiles = (nt + 15) >> 4;
  // software pipeline: the train rows of tile t + 2 are on their way from memory and those of tile t + 1 on their way
  // through the table while tile t is multiplied
  auto load_rows = [&](int t) {
    const uint32_t off = min((uint32_t)(16 * t + c), (uint32_t)(cap - 1)) * 32u + 8u * (uint32_t)g;  // (cap * 32 < 2^21)
    return *reinterpret_cast<const uint2*>(td + off);
  };
  v4i bx[4], by[4];
  {
    const uint2 w0 = load_rows(0);
    expand8(lut, lane_off, w0.x, w0.y, bx);
  }
  // train rows are requested FOUR tiles ahead (a ring of four 8-byte registers pairs): with two waves per SIMD nothing
  // else hides an L2 round trip.  Rows are clamped to the frame, a load past the last tile is harmless and unused.
  uint2 w0 = load_rows(1), w1 = load_rows(2), w2 = load_rows(3), w3 = load_rows(4);
  // The loop is software-pipelined ACROSS tiles and the issue order is pinned with sched_barrier: slot s of a tile is one
  // MFMA of THIS tile (chunk-major, so consecutive MFMAs never depend on each other) followed by the two VALU
  // instructions that fold one accumulator register of the PREVIOUS tile into best / second best, plus one eighth of
  // the table look-ups that expand the NEXT tile.  The VALU work of a wave then sits in the shadow of its own MFMAs
  // (an MFMA occupies the matrix core for about 20 clocks and the issue port for 4).
  v4i acc0[QT], acc1[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) acc1[qt] = v4i{-1, -1, -1, -1};  // "previous tile" of tile 0: keys that change nothing
  auto fold = [&](const v4i (&acc)[QT], int k) {
    const int qt = k >> 2, r = k & 3;
    const uint32_t key = (uint32_t)acc[qt][r];
    b2[qt][r] = umed3m(key, b1[qt][r], b2[qt][r]);
    // med3 first and the min not shared with the one inside the med3 pattern (the empty asm hides the equality): b1 is
    // then updated in place; otherwise every tile starts with 4 QT register copies of the loop-carried b1
    uint32_t old1 = b1[qt][r];
    asm("" : "+v"(old1));
    __builtin_amdgcn_sched_barrier(0);
    b1[qt][r] = min(old1, key);
  };
  auto tile = [&](int t, uint2& wring, const v4i (&cur)[4], v4i (&nxt)[4], v4i (&acc)[QT], const v4i (&prev)[QT]) {
    const uint2 w = wring;  // train rows of tile t + 1, requested four tiles ago
    wring = load_rows(t + 5);
    // train rows past the count start from a key above every real one
    const int j = 16 * t + c;
    const uint32_t kb = (j < nt ? (256u << 12) : kInvalidKey) + (uint32_t)t;
    const v4i cinit = {(int)kb, (int)kb, (int)kb, (int)kb};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const int s = m * QT + qt;
        acc[qt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[qt][m], cur[m], m == 0 ? cinit : acc[qt], 0, 0, 0);
        fold(prev, s);
        if (s < 8) {  // byte s of the lane's 8 bytes of the next tile -> half an operand chunk
          const uint32_t word = s < 4 ? w.x : w.y;
          const uint32_t sh = (s & 3) == 0 ? word << 8 : (s & 3) == 1 ? word : (s & 3) == 2 ? word >> 8 : word >> 16;
          const uint2 e = *reinterpret_cast<const uint2*>(lut + ((sh & 0xFF00u) | lane_off));
          nxt[s >> 1][2 * (s & 1)] = (int)e.x;
          nxt[s >> 1][2 * (s & 1) + 1] = (int)e.y;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // the tile count is rounded up to a multiple of four so that the body stays branch-free (the loads of the ring are
  // then waited for with vmcnt(3), not vmcnt(0)); the up to three extra tiles hold only rows past the count
  for (int t = 0; t < n_tiles; t += 4) {
    tile(t, w0, bx, by, acc0, acc1);
    tile(t + 1, w1, by, bx, acc1, acc0);
    tile(t + 2, w2, bx, by, acc0, acc1);
    tile(t + 3, w3, by, bx, acc1, acc0);
  }
  // the last tile's accumulators are still unfolded
#pragma unroll
  for (int k = 0; k < 4 * QT; ++k) fold(acc1, k);
  // ---- (key << 4) | c = (distance' << 16) | j; merge over the 16 lanes of each DPP row (the columns), then lane c == 0
  // of row g writes query rows 4 g + r
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
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

template <int QT, int WAVES>
gh_status launch_variant(gh_ctx* ctx, const uint8_t* desc_dev, const int32_t* counts_dev, int cap, const int32_t* pair_q_dev,
                         const int32_t* pair_t_dev, int npairs, int32_t* idx1_dev, uint16_t* d1_dev, uint16_t* d2_dev) {
  // pairs are walked by persistent workgroups (one per CU fits: 64 KB of LDS, 2 waves per SIMD); four times as many
  // workgroups as CUs so that the hardware's dynamic dispatch evens out ragged pairs
  const int gx = gh_div_up(cap, 16 * QT * WAVES);
  int gy = (4 * (ctx->cu_count > 0 ? ctx->cu_count : 256) + gx - 1) / gx;
  gy = gy < npairs ? gy : npairs;
  GH_LAUNCH(ctx, "bf_match_pairs_mfma", (bf_match_pairs_mfma_kernel<QT, WAVES>), dim3(gx, gy), dim3(64 * WAVES), 0, desc_dev,
            counts_dev, cap, pair_q_dev, pair_t_dev, npairs, idx1_dev, d1_dev, d2_dev);
  return GH_OK;
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
  GH_CHECK_ARG(ctx, ((uintptr_t)desc_dev & 7) == 0);
  // 4 query tiles per wave x 8 waves: 64 + 32 + 32 + 32 registers of queries / best pairs / train tiles / accumulators
  // leave two waves per SIMD; 6 and 8 query tiles amortise the per-tile work further but spill (measured slower)
  return launch_variant<4, 8>(ctx, desc_dev, counts_dev, cap, pair_q_dev, pair_t_dev, npairs, idx1_dev, d1_dev, d2_dev);
}
