// ORB front end for gfx950: integer pyramid -> FAST-9/16 score + NMS + per-cell lists -> per-level
// deterministic top-n selection -> intensity-centroid orientation + 7x7 integer Gaussian + 256-bit
// steered BRIEF.  Bit-exact with oracle/orb_oracle.c, whose header is the step-by-step spec.
//
// What it feeds in the reference (there is no ORB code in the GSLAM tree; only the output types):
//   GSLAM/core/Map.h:122-195  KeyPoint (28 B)      == gh_keypoint
//   GSLAM/core/Map.h:309-321  MapFrame::setKeyPoints(keypoints, N x 32 8UC1 descriptor GImage)
//
// CDNA4 mapping.  Everything is HBM-bound byte/integer work, batched over frames (grid.z = frame) so a
// launch carries >> 256 workgroups:
//   resize      1 thread = 8 x 4 output pixels from one 16-byte load per source row
//   fast_cells  256 threads = 64x64 px = 2x2 cells; 72x96 B tile + 66x72 B score tile in LDS; each
//               wave then owns one 32x32 cell: __ballot prefix compaction keeps raster order, so the
//               per-cell candidate lists are deterministic without atomics or sorting
//   select      one workgroup per (level, frame): LDS histogram over (rank, score) finds the quota
//               cut-off, block scans give the output slots - no sort, no float
//   describe    one wave per keypoint: 33x33 patch in LDS, wave-reduced integer moments, separable
//               integer blur in LDS, 4 x __ballot packs the 256 test bits
#include <stdlib.h>

#include "common.h"
#include "orb_quadtree.h"
#include "../../include/gslam_orb_tables.h"

namespace {

constexpr int kEdge = GH_ORB_EDGE;         // 19
constexpr int kCell = GH_ORB_CELL;         // 32
constexpr int kCap = GH_ORB_CELL_CAP;      // 32
constexpr int kMaxL = GH_ORB_MAX_LEVELS;   // 8
constexpr int kHistBins = kCap * 256;      // (rank, 255 - score)
constexpr int kCellRec = 8;                // dwords per compact cell record: count + the first 7 entries

// Test-only branch census (gh_orb_plan_debug_counters): which rarely taken paths an extraction went through.
enum {
  kDbgCells = 0,         // cells processed by fast_cells
  kDbgDenseCells,        // cells with more than 64 scored pixels (the list-based branch)
  kDbgOverflowCells,     // cells that wrote entries 7.. to their overflow slot
  kDbgCapCells,          // cells that kept exactly kCap entries
  kDbgRankDropped,       // cells with more than kCap candidates: ranks >= kCap dropped
  kDbgStrongSilenced,    // cells where a strong corner removed weaker candidates
  kDbgMaxQueue,          // longest pass-1 queue of a tile
  kDbgMaxNz,             // most scored pixels in one cell
  kDbgSelCut,            // (level, frame) selections where the quota cut something off
  kDbgSelTieSplit,       // ... and the cut-off bin was taken only in part
  kDbgSelOverflowCells,  // cells whose overflow entries orb_select visited (output pass)
  kDbgSelStreamed,       // select launches of the streaming (non-cached) variant
  kDbgUnusedSlots,       // zero-filled output rows
  kDbgStarvedLevels,     // (level, frame) selections with fewer candidates than quota
  kDbgWeakCells,         // cells with candidates but no corner above the initial threshold (a two-phase detector's second pass)
  kDbgResizePasses,      // wave-level passes through the fused resize body (one per wave with at least one item, per item block)
  kDbgCount = 16
};

struct DevTables {
  const int8_t* pattern;        // table mode: [30][256] offset pairs (upload_pattern)
  const int32_t* dir;           // [30][2]
  const int8_t* base_pattern;   // continuous steering: the unrotated tests [256][4] int8
};

// XCD-aware work mapping.  Workgroup b is observed to run on XCD b % 8 (each XCD has a private 4 MiB L2);
// this hands every XCD one contiguous strip of the tile list so that spatial neighbours (tile halos,
// overlapping keypoint patches) meet in the same L2.  Placement only affects speed, never results.
__device__ __forceinline__ int xcd_strip_tile(int lin, int total) {
  const int chunk = (total + 7) >> 3;
  return (lin & 7) * chunk + (lin >> 3);
}

// ------------------------------------------------------------------------------------------------
// level-0 staging copy (only when the caller's buffer is not dword friendly)
__global__ void copy_rows_kernel(const uint8_t* __restrict__ src, size_t src_frame_stride, int src_pitch,
                                 uint8_t* __restrict__ dst, size_t dst_frame_stride, int dst_pitch, int w, int h) {
  int x = blockIdx.x * blockDim.x + threadIdx.x;
  int y = blockIdx.y;
  if (x >= w) return;
  dst[(size_t)blockIdx.z * dst_frame_stride + (size_t)y * dst_pitch + x] =
      src[(size_t)blockIdx.z * src_frame_stride + (size_t)y * src_pitch + x];
}

// ------------------------------------------------------------------------------------------------
// bilinear 1.2x downscale, 11-bit fixed point (oracle step 1).  xtab/ytab: idx << 16 | frac.
// The kernel is bound by the NUMBER of vector-memory instructions, not by bytes or VALU (measured: 3 dword loads
// per source row -> one dwordx3 load gave -27 %), so a thread produces 8 x 4 outputs from ONE 16-byte load per
// source row: the 8 outputs of a row read source columns sx0 .. sx0 + 10 (scale 1.2) and the window starts at the
// dword that holds sx0.  Threads are numbered row-group-major over ceil(wd / 8) column groups so that waves stay
// full on the narrow levels.
constexpr int kResizeRows = 4;  // output rows per thread
constexpr int kResizeCols = 8;  // output columns per thread

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));  // 16 bytes at 4-byte alignment: ONE dwordx4 load

// a * b + c on the 24-bit multiplier (the compiler turns __umul24(a, b) + __umul24(c, d) + k into two multiplies and a
// three-input add: one instruction more per output pixel)
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// Per output column of a level, precomputed at plan creation (it depends on the column only): the v_perm selector that
// puts the column's two horizontal taps into the 16-bit halves of a dword, relative to the 16-byte window of its
// 8-column group, and the weight pair {2048 - fx, fx} -- the horizontal lerp of a source row is then ONE v_perm and ONE
// v_dot2_u32_u16 per output pixel with no per-item set-up arithmetic (it was ~7 VALU per column per item).
struct ResizeTabs {
  const uint32_t* xtab;  // idx << 16 | frac per output column, padded to 8 (only the group's first entry is read)
  const uint32_t* xsel;  // perm selector per output column
  const uint32_t* xwgt;  // fx << 16 | (2048 - fx)
  const uint32_t* ytab;  // idx << 16 | frac per output row
  // fused MFMA resize (resize_tile_mfma): per 8-column output group g the A operand of v_mfma_f32_16x16x32_f16 for the 16 output
  // columns 8 g .. 8 g + 15 -- A[m][k] = the weight of source column mcw[g] + k for output column 8 g + m (2048 - fx at the left
  // tap, fx at the right one, f16) -- as 64 lanes x 4 dwords, and the window start mcw[g] = sx(8 g) & ~7.  Null: not available.
  const uint32_t* mtab = nullptr;
  const uint32_t* mcw = nullptr;
};

// One work item: output columns x8 .. x8 + 7, output rows y0 .. y_end - 1 (at most kResizeRows of them) of one frame.
// s / d: the frame's source level and destination level (uniform per workgroup: all addressing is a scalar base plus a
// 32-bit lane offset).  ytab may be read at any (unaligned) y0.
__device__ __forceinline__ void resize_item(const LevelView& src, const uint8_t* __restrict__ s, uint8_t* __restrict__ d,
                                            int dst_pitch, const ResizeTabs& tb, int x8, int y0, int y_end, bool guard_frame) {
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  const int sx0 = (int)(tb.xtab[x8] >> 16);
  const uint32_t al = (uint32_t)(sx0 & 3);
  const uint32_t d0 = (uint32_t)sx0 & ~3u;  // byte offset of the dword that holds sx0
  const uint4 sa = *reinterpret_cast<const uint4*>(tb.xsel + x8), sb = *reinterpret_cast<const uint4*>(tb.xsel + x8 + 4);
  const uint4 wa = *reinterpret_cast<const uint4*>(tb.xwgt + x8), wb = *reinterpret_cast<const uint4*>(tb.xwgt + x8 + 4);
  const uint32_t selp[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
  const uint32_t wx[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
  // the 16-byte window may run up to 15 bytes past the end of a source row: harmless inside the buffer (next row,
  // next frame, or the slab's tail pad), but the caller's level-0 buffer has no pad after its last row
  const bool guard = guard_frame && (int)d0 + 16 > src.pitch;
  const uint32_t last_dw = (uint32_t)src.pitch - 4u;
  const uint32_t pitch = (uint32_t)src.pitch;
#pragma unroll
  for (int rr = 0; rr < kResizeRows; ++rr) {
    const int y = y0 + rr;
    if (y >= y_end) break;
    const uint32_t ty = tb.ytab[y];
    const uint32_t sy = ty >> 16;
    const uint32_t fy = ty & 0xFFFFu;
    const uint32_t sy1 = sy + 1 < (uint32_t)src.h ? sy + 1 : (uint32_t)src.h - 1;
    // 32-bit offsets inside the frame (a level is < 4 GiB); rows and pitch are < 2^24: the 24-bit multiplier is full rate, a
    // 32-bit v_mul_lo_u32 quarter rate (8 of them per item were 1 % of orb_fast_cells)
    const uint32_t o0 = __umul24(sy, pitch), o1 = __umul24(sy1, pitch);
    u32x4_a4 u, v;
    if (guard && sy1 == (uint32_t)src.h - 1) {
      auto ld = [&](uint32_t row, uint32_t k) { return *reinterpret_cast<const uint32_t*>(s + row + min(d0 + 4u * k, last_dw)); };
      u = u32x4_a4{ld(o0, 0), ld(o0, 1), ld(o0, 2), ld(o0, 3)};
      v = u32x4_a4{ld(o1, 0), ld(o1, 1), ld(o1, 2), ld(o1, 3)};
    } else {
      u = *reinterpret_cast<const u32x4_a4*>(s + (o0 + d0));
      v = *reinterpret_cast<const u32x4_a4*>(s + (o1 + d0));
    }
    // 12-byte windows that start exactly at sx0
    const uint32_t uw[3] = {__builtin_amdgcn_alignbyte(u.y, u.x, al), __builtin_amdgcn_alignbyte(u.z, u.y, al),
                            __builtin_amdgcn_alignbyte(u.w, u.z, al)};
    const uint32_t vw[3] = {__builtin_amdgcn_alignbyte(v.y, v.x, al), __builtin_amdgcn_alignbyte(v.z, v.y, al),
                            __builtin_amdgcn_alignbyte(v.w, v.z, al)};
    // vertical weights x4: the 2^22 weight total becomes 2^24, so the rounded result is the TOP BYTE of the sum
    // ((4 v + 2^23) >> 24 == (v + 2^21) >> 22, max 255 * 2^24 + 2^23 < 2^32): two v_mad_u32_u24 per pixel
    const uint32_t wy0 = 4u * (2048u - fy), wy1 = 4u * fy;
    uint32_t r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int wsel = i >= 4 ? 1 : 0;
      const u16x2 pu = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(uw[wsel + 1], uw[wsel], selp[i]));
      const u16x2 pv = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(vw[wsel + 1], vw[wsel], selp[i]));
      const u16x2 w2 = __builtin_bit_cast(u16x2, wx[i]);
      const uint32_t h0 = __builtin_amdgcn_udot2(pu, w2, 0u, false);  // a00 (2048 - fx) + a01 fx  (< 2^20)
      const uint32_t h1 = __builtin_amdgcn_udot2(pv, w2, 0u, false);
      r[i] = mad_u24(h0, wy0, mad_u24(h1, wy1, 1u << 23));
    }
    // top bytes of r[0..7] -> 8 output bytes: one v_perm per pixel pair, one v_perm per dword
    uint2 packed;
    packed.x = __builtin_amdgcn_perm(__builtin_amdgcn_perm(r[3], r[2], 0x0c0c0703u), __builtin_amdgcn_perm(r[1], r[0], 0x0c0c0703u), 0x05040100u);
    packed.y = __builtin_amdgcn_perm(__builtin_amdgcn_perm(r[7], r[6], 0x0c0c0703u), __builtin_amdgcn_perm(r[5], r[4], 0x0c0c0703u), 0x05040100u);
    // dst pitch is a multiple of 64 and the pad bytes are ours: always a full 8-byte store
    *reinterpret_cast<uint2*>(d + (__umul24((uint32_t)y, (uint32_t)dst_pitch) + (uint32_t)x8)) = packed;
  }
}

__global__ __launch_bounds__(256) void resize_kernel(LevelView src, uint8_t* __restrict__ dst_base,
                                                     size_t dst_frame_stride, int dst_pitch, int wd, int hd,
                                                     ResizeTabs tb, int groups_per_row,
                                                     uint32_t groups_inv, int n_items, int tail_unsafe_frame) {
  const int item = blockIdx.x * 256 + threadIdx.x;
  if (item >= n_items) return;
  const int rg = (int)__umulhi((uint32_t)item, groups_inv);  // item / groups_per_row (exact: checked on the host)
  const int x8 = (item - rg * groups_per_row) * kResizeCols;
  const int y0 = rg * kResizeRows;
  resize_item(src, src.base + (size_t)blockIdx.y * src.frame_stride, dst_base + (size_t)blockIdx.y * dst_frame_stride, dst_pitch,
              tb, x8, y0, min(hd, y0 + kResizeRows), (int)blockIdx.y == tail_unsafe_frame);
}

// The next pyramid level produced from inside fast_cells (see there): which output 8-column groups / output rows of
// level l + 1 the workgroups of tile column bx / tile row by own.
struct NextLevel {
  uint8_t* dst_base;       // level l + 1, frame 0; nullptr = nothing to produce
  size_t dst_frame_stride;
  int dst_pitch, hd;
  ResizeTabs tb;           // of level l + 1
  const int32_t* gx0;      // [nbx + 1] first owned output group per tile column
  const int32_t* gy0;      // [nby + 1] first owned output row per tile row
  int tail_unsafe_frame;
  // tile id -> (frame, tile row, tile column) without integer divisions: q = __umulhi(n, inv) == n / d for every tile id of
  // the launch (checked on the host: magic_div); 0 = divide (the one-launch-for-all-levels path of small calls)
  uint32_t tiles_inv = 0, nbx_inv = 0;
  // PLANE variant of the kernel (the quadtree mode's score plane, orb_quadtree.hip): S of this level, pixel (y, x) of a frame at
  // plane[y * plane_pitch + x + kQtPlaneX]
  uint8_t* plane = nullptr;
  size_t plane_frame_stride = 0;
  int plane_pitch = 0;
};

// inv with __umulhi(n, inv) == n / d for every n <= n_max, or 0 when no such 32-bit constant is guaranteed
inline uint32_t magic_div(uint32_t d, uint32_t n_max) {
  if (d == 0) return 0;
  if (d == 1) return 0;  // (n * 2^32 does not fit: let the kernel divide)
  const uint64_t inv = ((1ull << 32) + d - 1) / d;   // ceil(2^32 / d)
  const uint64_t excess = inv * d - (1ull << 32);    // < d
  return (inv < (1ull << 32) && (uint64_t)n_max * excess < (1ull << 32)) ? (uint32_t)inv : 0u;
}

// The next pyramid level of an INTERIOR tile from the image tile in LDS, horizontal taps on the matrix cores (VERDICT r4 item
// 1a; the same idea as orb_describe's blur, desc_blur_mfma).  For 16 output rows x 16 output columns:
//     D1[m][n] = sum_k A[m][k] B1[k][n],  A = the weights of ResizeTabs::mtab (rows = output columns), B1[k][n] = f16(1024 +
//     P[sy(y_n)][cw + k]) -- a byte OR 0x6400 --, B2 the same with row sy + 1;  C = 0.
// The bias adds 1024 * 2048: D = 2^21 + h with h = (2048 - fx) a + fx b < 2^19, a float in [2^21, 2^22) whose bits are
// 0x4A000000 | (h << 2) -- the low 24 bits are the integer 4 h (exact: every partial sum is an integer below 2^24).  The lane
// holds output row n = lane & 15, output columns 4 (lane >> 4) + j: the vertical lerp is two v_mad_u32_u24 per pixel straight on
// the accumulator bits, top byte = (h0 (2048 - fy) + h1 fy + 2^21) >> 22 as resize_item computes it, four bytes = one dword
// store.  16 VALU per 4 output pixels become ~7 (perms of the B operands shared by 4 columns) + two MFMAs.
// Interior: the tile's owned outputs read source rows oy .. oy + 64 and columns ax .. ax + 87 -- all inside the 72 x 96 tile,
// none clamped.  (row block, column block) pairs are dealt round-robin to the four waves; tile = the image tile in LDS.
struct ResizeOps {  // what resize_tile_mfma reads from global memory: one round trip, requested ahead of the tile stages
  uint4 aw[4];     // A operands (weights) of up to four column-block pairs
  int cwv[4];      // tile column of their K windows
  uint32_t ty;     // source row << 16 | fy of this lane's output row
};
__device__ __forceinline__ ResizeOps resize_tile_mfma_load(int ax, const ResizeTabs& tb, int g0, int ng, int r0, int r1) {
  const int lane = threadIdx.x & 63, n16 = lane & 15;
  const int rb = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ncb = (ng + 1) >> 1;  // <= 4
  const int yq = r0 + 16 * rb + n16, y = yq < r1 ? yq : r1 - 1;
  ResizeOps o;
  o.ty = tb.ytab[y];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int g = g0 + 2 * (cb < ncb ? cb : 0);
    o.aw[cb] = *reinterpret_cast<const uint4*>(tb.mtab + ((size_t)g * 64 + lane) * 4);
    o.cwv[cb] = (int)tb.mcw[g] - ax;  // tile column of the K window (a multiple of 8)
  }
  return o;
}
__device__ __forceinline__ void resize_tile_mfma(const uint8_t* tile, int tile_pitch, int oy, uint8_t* __restrict__ d,
                                                 int dst_pitch, const ResizeOps& ops, int g0, int ng, int r0, int r1,
                                                 uint32_t* __restrict__ dbg) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, n16 = lane & 15, q4 = lane >> 4;
  // wave w = row block w (an interior tile owns at most 64 output rows), all column blocks: ONE round of global loads up front
  // (the row's table entry, the column blocks' operands and window starts), then only LDS reads, MFMAs and the stores
  const int rb = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (r0 + 16 * rb >= r1) return;
  const int ncb = (ng + 1) >> 1;  // <= 4
  const int yq = r0 + 16 * rb + n16, y = yq < r1 ? yq : r1 - 1;
  const uint32_t ty = ops.ty;
  const uint4 (&aw)[4] = ops.aw;
  const int (&cwv)[4] = ops.cwv;
  const int R = (int)(ty >> 16) - oy;
  const uint32_t fy = ty & 0xFFFFu;
  const uint32_t wy0 = 2048u - fy, wy1 = fy, c23 = 1u << 23, k64 = 0x64646464u;
  const uint8_t* row1 = tile + R * tile_pitch + 8 * q4;
  const uint8_t* row2 = row1 + tile_pitch;
  const uint32_t rowoff = __umul24((uint32_t)y, (uint32_t)dst_pitch);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (2 * half >= ncb) break;
    uint32_t u[2][4], v[2][4];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int cb = 2 * half + e;
      const uint2 b1 = *reinterpret_cast<const uint2*>(row1 + cwv[cb]);
      const uint2 b2 = *reinterpret_cast<const uint2*>(row2 + cwv[cb]);
      uint4 f1, f2;
      f1.x = __builtin_amdgcn_perm(k64, b1.x, 0x04010400u);
      f1.y = __builtin_amdgcn_perm(k64, b1.x, 0x04030402u);
      f1.z = __builtin_amdgcn_perm(k64, b1.y, 0x04010400u);
      f1.w = __builtin_amdgcn_perm(k64, b1.y, 0x04030402u);
      f2.x = __builtin_amdgcn_perm(k64, b2.x, 0x04010400u);
      f2.y = __builtin_amdgcn_perm(k64, b2.x, 0x04030402u);
      f2.z = __builtin_amdgcn_perm(k64, b2.y, 0x04010400u);
      f2.w = __builtin_amdgcn_perm(k64, b2.y, 0x04030402u);
      const f32x4 c0 = {0.0f, 0.0f, 0.0f, 0.0f};
      const f16x8 af = __builtin_bit_cast(f16x8, aw[cb]);
      const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, __builtin_bit_cast(f16x8, f1), c0, 0, 0, 0);
      const f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, __builtin_bit_cast(f16x8, f2), c0, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u[e][j] = __float_as_uint(d1[j]);
        v[e][j] = __float_as_uint(d2[j]);
      }
    }
    uint32_t o[2][4];
    // (one asm statement for both column blocks that opens with its own wait states: the hazard recogniser does not look inside
    //  asm, and a VALU read of an MFMA result needs up to 18 of them -- see desc_blur_mfma)
    asm volatile(
        "s_nop 15\n\ts_nop 2\n\t"
        "v_mad_u32_u24 %0, %12, %25, %26\n\tv_mad_u32_u24 %1, %13, %25, %26\n\tv_mad_u32_u24 %2, %14, %25, %26\n\tv_mad_u32_u24 %3, %15, %25, %26\n\t"
        "v_mad_u32_u24 %4, %20, %25, %26\n\tv_mad_u32_u24 %5, %21, %25, %26\n\tv_mad_u32_u24 %6, %22, %25, %26\n\tv_mad_u32_u24 %7, %23, %25, %26\n\t"
        "v_mad_u32_u24 %0, %8, %24, %0\n\tv_mad_u32_u24 %1, %9, %24, %1\n\tv_mad_u32_u24 %2, %10, %24, %2\n\tv_mad_u32_u24 %3, %11, %24, %3\n\t"
        "v_mad_u32_u24 %4, %16, %24, %4\n\tv_mad_u32_u24 %5, %17, %24, %5\n\tv_mad_u32_u24 %6, %18, %24, %6\n\tv_mad_u32_u24 %7, %19, %24, %7"
        : "=&v"(o[0][0]), "=&v"(o[0][1]), "=&v"(o[0][2]), "=&v"(o[0][3]), "=&v"(o[1][0]), "=&v"(o[1][1]), "=&v"(o[1][2]), "=&v"(o[1][3])
        : "v"(u[0][0]), "v"(u[0][1]), "v"(u[0][2]), "v"(u[0][3]), "v"(v[0][0]), "v"(v[0][1]), "v"(v[0][2]), "v"(v[0][3]),
          "v"(u[1][0]), "v"(u[1][1]), "v"(u[1][2]), "v"(u[1][3]), "v"(v[1][0]), "v"(v[1][1]), "v"(v[1][2]), "v"(v[1][3]),
          "v"(wy0), "v"(wy1), "s"(c23));
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int cb = 2 * half + e;
      const uint32_t packed = __builtin_amdgcn_perm(__builtin_amdgcn_perm(o[e][3], o[e][2], 0x0c0c0703u),
                                                    __builtin_amdgcn_perm(o[e][1], o[e][0], 0x0c0c0703u), 0x05040100u);
      const int xo = 8 * (g0 + 2 * cb) + 4 * q4;
      if (cb < ncb && yq < r1 && xo < 8 * (g0 + ng)) *reinterpret_cast<uint32_t*>(d + (rowoff + (uint32_t)xo)) = packed;
    }
    if (dbg != nullptr && lane == 0) atomicAdd(&dbg[kDbgResizePasses], 1u);
  }
}

// Inclusive prefix sum of one int per lane over the wave: four DPP row shifts scan each 16-lane row, two row broadcasts
// (lane 15 -> next row on rows 1 / 3, lane 31 -> rows 2 / 3) carry the row totals: 6 VALU (five ballots and their
// masked popcounts were ~30).  Every lane of the wave must be active.
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31
  return v;
}

// ------------------------------------------------------------------------------------------------
// FAST-9/16 corner score (oracle step 2): max over 9-arcs of min(ring - p) / min(p - ring).
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

// r[] = the 16 ring intensities, c = the centre.  min/max commute with the subtraction of c, so the arcs are
// evaluated on the raw intensities (no 16 differences) and the running best takes two arcs per v_max3 / v_min3:
//   score = max(0, max_i min(arc_i) - c, c - min_i max(arc_i))
__device__ __forceinline__ int fast_score16(const int (&r)[16], int c) {
  int lo3[16], hi3[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    lo3[i] = min3i(r[i], r[(i + 1) & 15], r[(i + 2) & 15]);
    hi3[i] = max3i(r[i], r[(i + 1) & 15], r[(i + 2) & 15]);
  }
  int best_lo = 0, best_hi = 255;
#pragma unroll
  for (int i = 0; i < 16; i += 2) {
    const int lo9a = min3i(lo3[i], lo3[(i + 3) & 15], lo3[(i + 6) & 15]);
    const int lo9b = min3i(lo3[i + 1], lo3[(i + 4) & 15], lo3[(i + 7) & 15]);
    const int hi9a = max3i(hi3[i], hi3[(i + 3) & 15], hi3[(i + 6) & 15]);
    const int hi9b = max3i(hi3[i + 1], hi3[(i + 4) & 15], hi3[(i + 7) & 15]);
    best_lo = max3i(best_lo, lo9a, lo9b);
    best_hi = min3i(best_hi, hi9a, hi9b);
  }
  return max3i(0, best_lo - c, c - best_hi);
}

// The same score with BOTH polarities in one register: v = r | (255 - r) << 16, two 16-bit lanes whose bit patterns are
// fp16 DENORMALS (0 .. 255 -> exponent field 0), which order exactly as the integers do; the kernels run with fp16
// denormals preserved (amdhsa_float_denorm_mode_16_64 = 3).  min over an arc of the high lane is 255 - max over the arc of
// r, so ONE gfx950 v_pk_minimum3_f16 does the work of a v_min3_u32 and a v_max3_u32, and the running best of both
// polarities is a maximum: 16 packs + 16 + 16 + 8 three-input packed ops + 5 instead of 32 + 32 + 16 + 3 (bit-identical:
// every value is an exact small integer, no rounding anywhere).
__device__ __forceinline__ uint32_t pk_min3_f16(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ uint32_t pk_max3_f16(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ int fast_score16_pk(const int (&r)[16], int c) {
  uint32_t v[16], m3[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    // r - 65536 r + 255 * 65536 = r | (255 - r) << 16
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(v[i]) : "v"(r[i]), "v"(-65535), "v"(255 << 16));
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) m3[i] = pk_min3_f16(v[i], v[(i + 1) & 15], v[(i + 2) & 15]);
  uint32_t best = 0u;  // low lane: max(0, max_i min(arc_i)); high lane: 255 - min(255, min_i max(arc_i))
#pragma unroll
  for (int i = 0; i < 16; i += 2) {
    const uint32_t a = pk_min3_f16(m3[i], m3[(i + 3) & 15], m3[(i + 6) & 15]);
    const uint32_t b = pk_min3_f16(m3[i + 1], m3[(i + 4) & 15], m3[(i + 7) & 15]);
    best = pk_max3_f16(best, a, b);
  }
  const int best_lo = (int)(best & 0xFFFFu), hi_c = (int)(best >> 16);
  return max3i(0, best_lo - c, c - 255 + hi_c);
}

#ifndef GH_FAST_WAVES
#define GH_FAST_WAVES 8
#endif
// measuring builds only (-DGH_ORB_PHASES, tools/r6_phases.py): wall-clock time (100 MHz) thread 0 of every workgroup spends between the
// barriers of a tile, summed over the launch -- [0] start -> tile in LDS, [1] -> pass 1 done, [2] -> pass 2 done, [3] -> cell of wave 0
// done, [4] (tile loop) -> all waves done, [8] tiles
#ifdef GH_ORB_PHASES
__device__ unsigned int* g_orb_phase_buf;  // [slots][8] ticks per phase, one slot per tile (plain stores: no contended atomics)
__shared__ unsigned long long ph_t;
__shared__ unsigned int ph_slot;
#define GH_PHASE_START() do { if (threadIdx.x == 0) ph_t = wall_clock64(); } while (0)
#define GH_PHASE(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); g_orb_phase_buf[(size_t)ph_slot * 8 + (k)] = (unsigned int)(now_ - ph_t); ph_t = now_; } } while (0)
#else
#define GH_PHASE_START()
#define GH_PHASE(k)
#endif
// timing experiments only (docs/notes_r06.md: what each stage of orb_fast_cells costs; wrong results): bit 0 no pass 2, bit 1 no pass 1,
// bit 2 no cell stage, bit 3 no next pyramid level
#ifndef GH_ORB_WHATIF
#define GH_ORB_WHATIF 0
#endif
constexpr int kTileW = 96;   // bytes per LDS tile row (6 x 16 B: 16-byte aligned window that covers x0-4 .. x0+67), 72 rows
constexpr int kTileH = 72;
constexpr int kScoreH = 66;   // score window: 64x64 region + 1 px NMS halo
constexpr int kScoreOff = 3;  // window col sx is stored at byte sx + 3 so that both cells of a row start dword aligned
constexpr int kScoreWPk = 72;  // row pitch (bytes) of the packed-16-bit pass 1; the SWAR pass 1 uses 68 (see fast_cells_tile)

// One 64 x 64 tile (2 x 2 cells) of one level of one frame, by one 256-thread workgroup; tile_id in [0, nbx nby n_frames).
// PK: arc scores through fast_score16_pk (GSLAM_HIP_ORB_PKSCORE, decided per plan).
// P1: formulation of pass 1 (GSLAM_HIP_ORB_PASS1, decided per plan) -- 0 = packed 16-bit min / max (rounds 2-3),
//     1 = SWAR on 16-bit fields (full-rate and / or / sub / v_bitop3), fields split in registers.  (A variant that read the
//     fields pre-split from LDS planes lost 22 %: 38 KB of LDS leave 4 workgroups per CU; profiles/orb_pass1_ab_r04.txt.)
// tile id -> (frame, tile row, tile column)
__device__ __forceinline__ void fast_tile_coords(const NextLevel& nx, int nbx, int nby, int tile_id, int& frame, int& bx, int& by) {
  if (nx.tiles_inv != 0u && nx.nbx_inv != 0u) {  // (two integer divisions were ~40 VALU per wave: 4 % of the kernel)
    frame = (int)__umulhi((uint32_t)tile_id, nx.tiles_inv);
    const int trem = tile_id - frame * (nbx * nby);
    by = (int)__umulhi((uint32_t)trem, nx.nbx_inv);
    bx = trem - by * nbx;
  } else {
    frame = tile_id / (nbx * nby);
    const int trem = tile_id - frame * (nbx * nby);
    bx = trem % nbx;
    by = trem / nbx;
  }
}
// 16-byte item i (< kTileH * 6) of the image tile of (frame, bx, by): LDS byte offset 16 i, global source clamped to the level
__device__ __forceinline__ const uint4* fast_tile_src(const LevelView& lv, int frame, int bx, int by, int i) {
  // i / 6 on the full-rate 24-bit multiplier: exact for i < 420 since 10923 / 65536 - 1 / 6 = 5e-6 (a division by a
  // constant costs a quarter-rate v_mul_hi)
  const int row = (int)(__umul24((uint32_t)i, 10923u) >> 16), c = i - row * 6;
  int gy = kEdge + 64 * by - 4 + row;
  gy = gy < 0 ? 0 : (gy > lv.h - 1 ? lv.h - 1 : gy);
  int gx = 64 * bx + 16 * c;
  gx = gx > lv.pitch - 16 ? lv.pitch - 16 : gx;
  // (rows and pitch are < 2^24, a level is < 4 GiB: 32-bit offset, full-rate multiply instead of a 64-bit v_mad_i64_i32)
  return reinterpret_cast<const uint4*>(lv.base + (size_t)frame * lv.frame_stride + (__umul24((uint32_t)gy, (uint32_t)lv.pitch) + (uint32_t)gx));
}

template <bool PK, int P1, bool PLANE = false>
__device__ __forceinline__ void fast_cells_tile(const LevelView& lv, int ncx, int ncy, int min_th, int ini_th,
                                                uint32_t* __restrict__ cell_cnt, uint32_t* __restrict__ cell_ent,
                                                int cells_per_frame, int cell_off, int n_frames, const NextLevel& nx,
                                                uint32_t* __restrict__ dbg, int tile_id) {
  // SWAR pass 1 covers a window row with 17 aligned dwords (window cols -2 .. 65) and the score tile is FLAT with 68 bytes
  // per row: byte 68 sy + 3 + sx, so (item, pixel k) lands at 4 item + 1 + k.  Cols -2 / -1 of a row share their bytes with
  // cols 66 / 67 of the row above; nothing ever reads them (NMS looks at cols 0 .. 65 only).
  constexpr int kScoreW = P1 != 0 ? 68 : kScoreWPk;
  constexpr int kScoreBytes = P1 != 0 ? kScoreH * 68 + 8 : kScoreH * kScoreWPk;
  constexpr int kQueueLen = P1 != 0 ? kScoreH * 68 : kScoreH * kScoreH;
  __shared__ __attribute__((aligned(16))) uint8_t tile[kTileH * kTileW];
  __shared__ __attribute__((aligned(16))) uint8_t score[kScoreBytes];
  // the dense-cell lists of the NMS stage live in the image tile, which is dead after pass 2 (P1 != 0: 20.4 KB of LDS per
  // workgroup -> 7-8 workgroups per CU instead of 6; the kernel is latency sensitive: profiles/orb_pass1_ab_r04.txt)
  __shared__ uint32_t lists_own[P1 == 0 ? 4 * 256 : 1];
  uint32_t(*lists)[256] = reinterpret_cast<uint32_t(*)[256]>(P1 == 0 ? reinterpret_cast<uint8_t*>(lists_own) : tile);
  static_assert(sizeof(tile) >= 4 * 256 * sizeof(uint32_t), "four 1 KB lists fit the tile");
  __shared__ uint16_t queue[kQueueLen];
  __shared__ int q_count;
  __shared__ uint16_t bit_pos[32];  // P1 != 0: score-tile offset of candidate bit b of a thread, relative to 4 tid
  GH_PHASE_START();

  const int tid = threadIdx.x;
  __builtin_assume(tid >= 0 && tid < 256);
  const int nbx = (ncx + 1) >> 1, nby = (ncy + 1) >> 1;
  int frame, bx, by;
  fast_tile_coords(nx, nbx, nby, tile_id, frame, bx, by);
#ifdef GH_ORB_PHASES
  if (threadIdx.x == 0) {
    const unsigned int slot_ = (unsigned int)(frame * cells_per_frame + cell_off + by * nbx + bx);
    ph_slot = slot_;
    g_orb_phase_buf[(size_t)slot_ * 8 + 6] = 1u;
  }
#endif
  static_assert(kTileW / 16 == 6 && kTileH * (kTileW / 16) <= 512 && kTileH * (kTileW / 16) > 256, "two 16-byte items per thread");
  if (tid == 0) q_count = 0;
  const int x0 = kEdge + 64 * bx, y0 = kEdge + 64 * by;  // region origin
  const int oy = y0 - 4;                       // tile origin row
  const int ax = 64 * bx;                      // 16-byte aligned tile origin column: x0 - 4 == ax + 15
  const uint8_t* img = lv.base + (size_t)frame * lv.frame_stride;

  // interior tile: the operands of the fused MFMA resize are requested FIRST, in front of the tile loads -- their round trip (L2)
  // runs under the tile's (HBM) instead of behind the first barrier (round 6: -2 % on the kernel)
  bool resized = false;
  ResizeOps rops;
  int rg0 = 0, rng = 0, rr0 = 0, rr1 = 0;
  if constexpr (P1 != 0) {
    if (!(GH_ORB_WHATIF & 8) && nx.dst_base != nullptr && nx.tb.mtab != nullptr && by > 0 && by < nby - 1 && bx < nbx - 1) {
      rg0 = nx.gx0[bx];
      rng = nx.gx0[bx + 1] - rg0;
      if (rng <= 8) {
        rr0 = nx.gy0[by];
        rr1 = nx.gy0[by + 1];
        rops = resize_tile_mfma_load(ax, nx.tb, rg0, rng, rr0, rr1);
        resized = true;
      }
    }
  }
  // 16 B per lane: rows of the level are 16-byte aligned (pitch % 16 == 0, checked by the launcher)
  for (int i = tid; i < kTileH * (kTileW / 16); i += 256) *reinterpret_cast<uint4*>(&tile[16 * i]) = *fast_tile_src(lv, frame, bx, by, i);
  if constexpr (P1 != 0) {
    // candidate bit b of a thread (see pass 1): b ^ 15 = 16 (k >> 1) + 2 trip + (k & 1) -> score offset 1024 trip + 1 + k
    if (tid < 32) {
      const int ix = tid ^ 15, k = (ix & 1) + 2 * (ix >> 4), trip = (ix >> 1) & 7;
      bit_pos[tid] = (uint16_t)(1024 * trip + 1 + k);
    }
  }
  __syncthreads();
  GH_PHASE(0);
  // The next pyramid level of an INTERIOR tile, now, out of the image tile (which the NMS lists overwrite after pass 2): h-taps
  // on MFMA.  Edge tiles (first / last tile row, last tile column: they also own what lies outside every tile) keep the VALU
  // path at the end of the kernel, which reads global memory.
  if constexpr (P1 != 0) {
    if (resized) resize_tile_mfma(tile, kTileW, oy, nx.dst_base + (size_t)frame * nx.dst_frame_stride, nx.dst_pitch, rops, rg0, rng, rr0, rr1, dbg);
  }

  // Scores for the 66x66 window (region + 1 px NMS halo); tile col of window col sx is sx + 18, row sy + 3.
  // Pass 1: cheap necessary condition on the 4 compass pixels (any 9-arc holds two ADJACENT compass
  // pixels) for every pixel; survivors are appended to an LDS queue so that pass 2 (the ~100-op arc
  // score) runs with all lanes busy instead of paying full price in every partially-hit wave.
  // The queue order is irrelevant: results land in score[] by position.
  // score tile cleared (incl. pad cols): 16 bytes per store where the size allows (two trips instead of five)
  if constexpr (kScoreBytes % 16 == 0) {
    for (int i = tid; i < kScoreBytes / 16; i += 256) reinterpret_cast<uint4*>(score)[i] = make_uint4(0u, 0u, 0u, 0u);
  } else {
    for (int i = tid; i < kScoreBytes / 4; i += 256) reinterpret_cast<uint32_t*>(score)[i] = 0u;
  }
  // One work item = one aligned tile dword = 4 horizontally adjacent pixels (tile cols 4m .. 4m+3, m = 4..21,
  // i.e. window cols -2 .. 69), evaluated with packed 16-bit math: 5 LDS dword reads and ~56 VALU per 4 px.
  // bright test: some adjacent compass pair both > c + t; dark: both < c - t.
  const int sx_lo = max(0, kEdge - (x0 - 1)), sx_hi = min(kScoreH, lv.w - kEdge - (x0 - 1));
  // tile-uniform: every window pixel lies in the valid region [kEdge, dim - kEdge) -> constant trim masks
  const bool interior = sx_lo == 0 && sx_hi == kScoreH && y0 - 1 >= kEdge && y0 - 1 + kScoreH <= lv.h - kEdge;
  if constexpr (P1 != 0) {
    // SWAR pass 1.  Every byte sits zero-extended in a 16-bit field: E(w) = {b0, b2}, O(w) = {b1, b3} of a tile dword.
    // With A = c + t + BIAS per field, A - x keeps its BIAS bit iff x <= c + t (x is NOT brighter), and with
    // D = c - t - 1 + BIAS, D - x keeps it iff x < c - t (x IS darker); fields never borrow from each other (|c +- t - x| < 2^10).
    // bright = (b_u | b_d) & (b_l | b_r), dark likewise, folded with and / or / v_bitop3: 4 px per instruction instead of 2,
    // and the compare / fold instructions are in the full-rate ALU class (profiles/issue_probe_r03.txt).  BIAS = 2^15 for the even pixels of the
    // dword and 2^14 for the odd ones, so that the four flags come out at bits 15, 14, 31, 30 (pixels 0, 1, 2, 3)
    // without any shifting; trip t parks them 2 t bits lower.  No trimming here: the two extra pixels of a row (window
    // cols -2, -1) and, in border tiles, pixels outside the valid region are dropped (or harmlessly scored) in pass 2.
    constexpr int kItemsPerRow = 17, kItems = kScoreH * kItemsPerRow;
    constexpr int kTrips = (kItems + 255) / 256;  // 5
    static_assert(2 * kTrips <= 14 && kScoreW == 4 * kItemsPerRow, "candidate bits of all trips share one dword");
    constexpr uint32_t kF = 0x00FF00FFu;
    const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane(min(max(min_th, 0), 255));  // (scalar: the constants below are s_mul)
    const uint32_t kAe = (0x8000u + t) * 0x10001u, kDe = (0x8000u - t - 1u) * 0x10001u;
    const uint32_t kAo = (0x4000u + t) * 0x10001u, kDo = (0x4000u - t - 1u) * 0x10001u;
    const uint32_t* tile32 = reinterpret_cast<const uint32_t*>(tile);
    const uint32_t tid3856 = (uint32_t)tid * 3856u;  // (item * 3856) >> 16 == item / 17 for item < 3855
    uint32_t allbits = 0;
#pragma unroll
    for (int trip = 0; trip < ((GH_ORB_WHATIF & 2) ? 0 : kTrips); ++trip) {
      const int item = 256 * trip + tid;
      if (item < kItems) {
        const uint32_t sy = (tid3856 + 3856u * 256u * (uint32_t)trip) >> 16;
        const uint32_t ix = (uint32_t)item + 7u * sy + (3 * (kTileW / 4) + 4);  // (sy + 3) * 24 + m, m = item - 17 sy + 4
        const uint32_t wc = tile32[ix], wl = tile32[ix - 1], wr = tile32[ix + 1];
        const uint32_t wu = tile32[ix - 3 * (kTileW / 4)], wd = tile32[ix + 3 * (kTileW / 4)];
        // E(w) by one v_and, O(w) by one v_perm (bytes 1, 3 -> the two fields)
        constexpr uint32_t kOdd = 0x0c030c01u;
        const uint2 C{wc & kF, __builtin_amdgcn_perm(0u, wc, kOdd)}, U{wu & kF, __builtin_amdgcn_perm(0u, wu, kOdd)},
            D{wd & kF, __builtin_amdgcn_perm(0u, wd, kOdd)};
        // even pixels (cols 4m, 4m+2): left = cols 4m-3, 4m-1 = O(wl); right = cols 4m+3, 4m+5 = {wc.b3, wr.b1}
        // odd pixels (cols 4m+1, 4m+3): left = cols 4m-2, 4m = {wl.b2, wc.b0}; right = cols 4m+4, 4m+6 = E(wr)
        // (v_perm_b32 D, S0, S1: selector bytes 0-3 address S1, 4-7 address S0, 0x0c = zero)
        const uint32_t le = __builtin_amdgcn_perm(0u, wl, kOdd), re = __builtin_amdgcn_perm(wr, wc, 0x0c050c03u);
        const uint32_t lo = __builtin_amdgcn_perm(wc, wl, 0x0c040c02u), ro = wr & kF;
        auto half = [](uint32_t A, uint32_t Dk, uint32_t u, uint32_t d, uint32_t l, uint32_t r) {
          const uint32_t not_bright_lr = (A - l) & (A - r);
          const uint32_t bright = __builtin_amdgcn_bitop3_b32(A - u, A - d, not_bright_lr, 0x15);  // ~(a & b) & ~c
          const uint32_t dark_lr = (Dk - l) | (Dk - r);
          const uint32_t dark = __builtin_amdgcn_bitop3_b32(Dk - u, Dk - d, dark_lr, 0xA8);  // (a | b) & c
          return bright | dark;
        };
        const uint32_t ye = half(C.x + kAe, C.x + kDe, U.x, D.x, le, re);
        const uint32_t yo = half(C.y + kAo, C.y + kDo, U.y, D.y, lo, ro);
        const uint32_t w = __builtin_amdgcn_bitop3_b32(ye, yo, 0x80008000u, 0xE4);  // (a & c) | (b & ~c)
        allbits = __builtin_amdgcn_bitop3_b32(allbits, w >> (2 * trip), 0xC000C000u >> (2 * trip), 0xF8);  // a | (b & c)
      }
    }
    // ONE queue reservation per wave for all trips; an entry is (bit << 8 | tid), decoded in pass 2 where every lane is busy
    {
      const int cnt = __popc(allbits);
      const int incl = wave_incl_scan_i32(cnt);
      const int wave_total = __builtin_amdgcn_readlane(incl, 63);
      if (wave_total != 0) {
        int base = 0;
        if ((tid & 63) == 0) base = atomicAdd(&q_count, wave_total);
        base = __builtin_amdgcn_readfirstlane(base) + incl - cnt;
        while (allbits) {
          const int b = __ffs((int)allbits) - 1;
          queue[base++] = (uint16_t)((b << 8) | tid);
          allbits &= allbits - 1u;
        }
      }
    }
  } else {
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    const uint32_t* tile32 = reinterpret_cast<const uint32_t*>(tile);
    constexpr int kRowDw = kTileW / 4, kItemsPerRow = 18;
    const s16x2 T = {(short)min_th, (short)min_th};
    constexpr int kTrips = (kScoreH * kItemsPerRow + 255) / 256;  // 5
    static_assert(4 * kTrips <= 32 && kScoreW == 4 * kItemsPerRow, "candidate bits of all trips share one dword");
    uint32_t allbits = 0;  // bit 4 t + k: pixel k of this thread's dword in trip t is a candidate
#pragma unroll
    for (int trip = 0; trip < kTrips; ++trip) {
      const int item = 256 * trip + tid;
      uint32_t bits = 0;  // bit k: pixel k of the dword is a candidate
      int sy = 0, m = 4;
      if (item < kScoreH * kItemsPerRow) {
        sy = item / kItemsPerRow;
        m = item - sy * kItemsPerRow + 4;
        const int trow = (sy + 3) * kRowDw;
        const uint32_t wc = tile32[trow + m], wl = tile32[trow + m - 1], wr = tile32[trow + m + 1];
        const uint32_t wu = tile32[trow - 3 * kRowDw + m], wd = tile32[trow + 3 * kRowDw + m];
        const uint32_t left4 = __builtin_amdgcn_alignbyte(wc, wl, 1);   // cols 4m-3 .. 4m
        const uint32_t right4 = __builtin_amdgcn_alignbyte(wr, wc, 3);  // cols 4m+3 .. 4m+6
        uint32_t eh[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          // bytes (2h, 2h+1) -> two zero-extended 16-bit lanes
          const uint32_t sel = h == 0 ? 0x0c010c00u : 0x0c030c02u;
          const s16x2 c = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, wc, sel));
          const s16x2 pu = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, wu, sel));
          const s16x2 pd = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, wd, sel));
          const s16x2 pl = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, left4, sel));
          const s16x2 pr = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, right4, sel));
          // every adjacent compass pair holds one vertical (up/down) and one horizontal (left/right) pixel, so
          // "some adjacent pair both brighter than c + t" == min(max(up, down), max(left, right)) > c + t, evaluated on
          // the raw intensities (min / max commute with subtracting c: no per-pixel differences); likewise for darker
          const s16x2 bright = __builtin_elementwise_min(__builtin_elementwise_max(pu, pd), __builtin_elementwise_max(pl, pr));
          const s16x2 dark = __builtin_elementwise_max(__builtin_elementwise_min(pu, pd), __builtin_elementwise_min(pl, pr));
          // sign bits: (c + t) - bright < 0  <=>  bright > c + t ;  dark - (c - t) < 0  <=>  dark < c - t
          const uint32_t e = __builtin_bit_cast(uint32_t, (c + T) - bright) | __builtin_bit_cast(uint32_t, dark - (c - T));
          eh[h] = e;
        }
        // the four sign bits (bit 15 / 31 of eh[0], eh[1]) -> a nibble: high bytes picked by one v_perm, flags gathered
        // by one v_dot4_u32_u8 with weights 1, 2, 4, 8
        const uint32_t mask = __builtin_amdgcn_udot4((__builtin_amdgcn_perm(eh[1], eh[0], 0x07050301u) >> 7) & 0x01010101u,
                                                     0x08040201u, 0u, false);
        // keep only pixels inside the 66-wide window and the valid image region: window cols [sx_lo, sx_hi)
        if (interior) {
          // whole window valid: only the dwords that stick out of it are trimmed (m = 4: window cols -2, -1; m = 21: 66 .. 69)
          const uint32_t vm = m == 4 ? 0xCu : (m == 21 ? 0u : 0xFu);
          bits = mask & vm;
        } else {
          const int py = y0 - 1 + sy;
          const int first = 4 * m - 18;  // window col of pixel 0 of this dword
          uint32_t vm = 0xFu;
          if (first < sx_lo) vm = (0xFu << min(sx_lo - first, 4)) & 0xFu;
          if (first + 4 > sx_hi) vm &= 0xFu >> min(first + 4 - sx_hi, 4);
          if (py >= kEdge && py < lv.h - kEdge) bits = mask & vm;
        }
      }
      allbits |= bits << (4 * trip);
    }
    // ONE queue reservation per wave for all trips: per-lane count (0..20), wave prefix sum (DPP scan), then
    // each lane unpacks its bits.  Score position of (item, k) = kScoreW sy + kScoreOff + 4 m - 18 + k with
    // m = item - 18 sy + 4, which is 4 item + 1 + k: no division needed.
    {
      const int cnt = __popc(allbits);
      const int incl = wave_incl_scan_i32(cnt);
      const int wave_total = __builtin_amdgcn_readlane(incl, 63);
      if (wave_total != 0) {
        int base = 0;
        if ((tid & 63) == 0) base = atomicAdd(&q_count, wave_total);
        base = __builtin_amdgcn_readfirstlane(base) + incl - cnt;
        const int p0 = 4 * tid + 1;
        while (allbits) {
          const int b = __ffs((int)allbits) - 1;
          queue[base++] = (uint16_t)(p0 + ((b >> 2) << 10) + (b & 3));
          allbits &= allbits - 1u;
        }
      }
    }
  }
  __syncthreads();
  GH_PHASE(1);
  const int nq = (GH_ORB_WHATIF & 1) ? 0 : q_count;
  for (int i = tid; i < nq; i += 256) {
    int pos = queue[i];
    if constexpr (P1 != 0) pos = 4 * (pos & 255) + bit_pos[pos >> 8];
    // (flat 68-byte rows: position 68 sy + 1 + j holds window col j - 2, j = 0 .. 67)
    // P1: n / 68 on the full-rate 24-bit multiplier (exact for n < 68 * 72: 15421 / 2^20 - 1 / 68 = 7.6e-7; a signed division by
    // a constant is a quarter-rate v_mul_hi_i32 and three fix-up instructions)
    int sy;
    if constexpr (P1 != 0) {
      static_assert(kScoreW == 68, "the constant below divides by 68");
      sy = (int)(__umul24((uint32_t)(pos - 1), 15421u) >> 20);
    } else {
      sy = pos / kScoreW;
    }
    const int sx = pos - sy * kScoreW - kScoreOff;
    if constexpr (P1 != 0) {
      // border tiles: pass 1 did not trim -- pixels outside the valid region [kEdge, dim - kEdge) keep S = 0 (oracle step 2)
      if (!interior && (sx < sx_lo || sx >= sx_hi || y0 - 1 + sy < kEdge || y0 - 1 + sy >= lv.h - kEdge)) continue;
    }
    const uint8_t* p = &tile[(sy + 3) * kTileW + sx + 18];
    const int c = p[0];
    int r[16];
    r[0] = p[-3 * kTileW];
    r[1] = p[-3 * kTileW + 1];
    r[2] = p[-2 * kTileW + 2];
    r[3] = p[-1 * kTileW + 3];
    r[4] = p[3];
    r[5] = p[1 * kTileW + 3];
    r[6] = p[2 * kTileW + 2];
    r[7] = p[3 * kTileW + 1];
    r[8] = p[3 * kTileW];
    r[9] = p[3 * kTileW - 1];
    r[10] = p[2 * kTileW - 2];
    r[11] = p[1 * kTileW - 3];
    r[12] = p[-3];
    r[13] = p[-1 * kTileW - 3];
    r[14] = p[-2 * kTileW - 2];
    r[15] = p[-3 * kTileW - 1];
    const int s = PK ? fast_score16_pk(r, c) : fast_score16(r, c);
    if (s > min_th) score[pos] = (uint8_t)s;
  }
  __syncthreads();
  GH_PHASE(2);

  // (placed after the last block-wide barrier: a wave that is done here goes straight on to its cell, nobody waits)
  // The next pyramid level, fused: this workgroup resizes the part of level l + 1 whose source pixels it has just
  // pulled through its L1 / the XCD's L2 for the tile (edge tiles also take the image border).  The pipeline is VALU-issue
  // bound and a stand-alone resize pass is memory-instruction bound, so the bilinear arithmetic rides in this kernel's idle
  // memory slots and the level is read from HBM once instead of twice.  Same arithmetic as resize_kernel (resize_item).
  if (!(GH_ORB_WHATIF & 8) && nx.dst_base != nullptr && !resized) {
    const int g0 = nx.gx0[bx], ng = nx.gx0[bx + 1] - g0;
    const int r0 = nx.gy0[by], r1 = nx.gy0[by + 1];
    const int nq = (r1 - r0 + kResizeRows - 1) / kResizeRows;
    const uint8_t* s = lv.base + (size_t)frame * lv.frame_stride;
    uint8_t* d = nx.dst_base + (size_t)frame * nx.dst_frame_stride;
    // lane -> (group g = tid & 7, row quad q = tid >> 3): a tile owns 6-7 groups x ~14 row quads, so one pass of 8 x 32
    // slots covers it with shifts and masks only (edge tiles that also take the image border loop over further blocks)
    for (int gb = 0; gb < ng; gb += 8) {
      const int g = gb + (tid & 7);
      for (int q = tid >> 3; q < nq; q += 32) {
        const int yy = r0 + kResizeRows * q;
        if (dbg != nullptr && __ffsll((unsigned long long)__ballot(true)) - 1 == (tid & 63)) atomicAdd(&dbg[kDbgResizePasses], 1u);
        if (g < ng)
          resize_item(lv, s, d, nx.dst_pitch, nx.tb, 8 * (g0 + g), yy, min(r1, yy + kResizeRows), frame == nx.tail_unsafe_frame);
      }
    }
  }

  if constexpr (PLANE) {
    // the quadtree mode wants S itself: the tile's 64 x 64 scores (zero outside the valid region) to the level's score plane,
    // 16 dwords per row (score byte 68 (r + 1) + 4 + c is dword aligned at c % 4 == 0; plane column x0 + kQtPlaneX + c starts a 64-byte
    // segment: a tile row is ONE aligned 64-byte write -- at x + 1 it straddled two, and the kernel was store bound)
    static_assert(P1 != 0, "the plane variant is written for the flat 68-byte score rows");
    uint8_t* pl = nx.plane + (size_t)frame * nx.plane_frame_stride;
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(score);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = tid + 256 * k, r = idx >> 4, cd = idx & 15;
      if (y0 + r < lv.h && x0 + kQtPlaneX + 4 * cd + 3 < nx.plane_pitch)
        *reinterpret_cast<uint32_t*>(pl + (__umul24((uint32_t)(y0 + r), (uint32_t)nx.plane_pitch) + (uint32_t)(x0 + kQtPlaneX + 4 * cd))) =
            s32[17 * (r + 1) + 1 + cd];
    }
    return;
  }
  // one wave per 32x32 cell
  const int wv = tid >> 6, lane = tid & 63;
  const int cx = 2 * bx + (wv & 1), cy = 2 * by + (wv >> 1);
  if (cx >= ncx || cy >= ncy) return;  // no block-wide sync below
  uint32_t* list = lists[wv];
  // 3x3 strict NMS over the cell.  Stage A compacts the few non-zero pixels of the cell, in raster order, into
  // list2; stage B tests only those against their 8 neighbours with every lane busy.  Both keep raster order.
  const int sy0 = 1 + 32 * (wv >> 1);
  const int col0 = kScoreOff + 1 + 32 * (wv & 1);  // byte column of the cell's first pixel: 4 or 36
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  uint16_t* list2 = queue + wv * 1024;  // the pass-1 queue is dead after pass 2: reuse it (row << 5 | col per entry)
  // Stage A: lane l scans 16 consecutive score bytes in raster order (cell row l >> 1, cols 16 (l & 1) ..), so lane
  // order IS raster order and one wave prefix sum of the per-lane non-zero counts (0..16, five ballots) places
  // every non-zero pixel; each lane then unpacks its own bits.
  int nz = 0;
  {
    const int row = lane >> 1, cb = 16 * (lane & 1);
    const uint32_t* sp32 = reinterpret_cast<const uint32_t*>(&score[(sy0 + row) * kScoreW + col0 + cb]);
    uint32_t nzb = 0;  // bit k: byte k of the 16 is non-zero
#pragma unroll
    for (int dwi = 0; dwi < 4; ++dwi) {
      const uint32_t w = sp32[dwi];
      // byte k -> 1 if non-zero, then the four flags are gathered into a nibble by one v_dot4_u32_u8 with weights
      // 1, 2, 4, 8 (a 32-bit multiply would be quarter rate)
      const uint32_t f = ((((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) >> 7) & 0x01010101u;
      nzb |= __builtin_amdgcn_udot4(f, 0x08040201u, 0u, false) << (4 * dwi);
    }
    const int cnt = __popc(nzb);
    const int incl = wave_incl_scan_i32(cnt);
    int pos = incl - cnt;
    nz = __builtin_amdgcn_readlane(incl, 63);
    const uint32_t rc0 = (uint32_t)((row << 5) | cb);
    while (nzb) {
      const int k = __ffs((int)nzb) - 1;
      list2[pos++] = (uint16_t)(rc0 + (uint32_t)k);
      nzb &= nzb - 1u;
    }
  }
  // (cy * ncx + cx < 2^24: the per-wave part on the 24-bit multiplier, the per-frame part is scalar)
  const size_t cell = (size_t)frame * cells_per_frame + cell_off + (size_t)(__umul24((uint32_t)cy, (uint32_t)ncx) + (uint32_t)cx);
  // cell record: {count, entries 0..6} in one 32-byte line of cell_cnt (a cell holds ~5 entries: orb_select reads ONE
  // 32-byte record per cell instead of a count plus a 128-byte slot); entries 7.. go to the cell's slot in cell_ent
  uint32_t* rec = cell_cnt + cell * kCellRec;
  uint32_t* ovf = cell_ent + cell * kCap;
  if (GH_ORB_WHATIF & 4) {
    if (lane == 0) rec[0] = 0u;
    return;
  }
  auto put_entry = [&](int idx, uint32_t v) {
    if (idx < kCellRec - 1) rec[1 + idx] = v;
    else ovf[idx] = v;
  };
  int kept = 0;
  if (nz <= 64) {
    // Common case (a cell holds ~25 scored pixels): one candidate per lane, everything stays in registers.  NMS
    // test, then the strong-corner filter and the rank-in-cell are evaluated over the ballot masks with v_readlane
    // broadcasts; lane order = raster order, so no list is written or re-read.
    bool ismax = false;
    uint32_t e = 0;
    int sv = 0;
    if (lane < nz) {
      const uint32_t rc = list2[lane];
      const uint8_t* sp = &score[(sy0 + (int)(rc >> 5)) * kScoreW + col0 + (int)(rc & 31u)];
      sv = sp[0];
      e = ((uint32_t)sv << 10) | rc;
      const int n0 = sp[-kScoreW - 1], n1 = sp[-kScoreW], n2 = sp[-kScoreW + 1], n3 = sp[-1], n4 = sp[1],
                n5 = sp[kScoreW - 1], n6 = sp[kScoreW], n7 = sp[kScoreW + 1];
      ismax = max3i(max3i(n0, n1, n2), max3i(n3, n4, n5), max(n6, n7)) < sv;
    }
    const uint64_t strong_mask = __ballot(ismax && sv > ini_th);
    const uint64_t cand = strong_mask != 0ull ? strong_mask : __ballot(ismax);  // a strong corner silences the weak ones
    // rank = candidates that come first by (score desc, raster asc).  With the raster bits of e inverted that order is ONE unsigned
    // comparison (lane order is raster order, rc is unique): v_readlane + v_cmp + v_addc per candidate instead of three compares
    const uint32_t key = e ^ 0x3FFu;
    int rank = 0;
#ifndef GH_ORB_WHATIF_NORANK  // (timing experiment only: what the in-cell ranking costs -- docs/notes_r06.md)
    for (uint64_t mm = cand; mm != 0ull; mm &= mm - 1ull) {
      const int j = __ffsll((unsigned long long)mm) - 1;
      rank += (uint32_t)__builtin_amdgcn_readlane((int)key, j) > key ? 1 : 0;
    }
#endif
    const bool keep = ((cand >> lane) & 1ull) != 0ull && rank < kCap;
    const uint64_t m = __ballot(keep);
    if (keep) put_entry(__popcll(m & lt_mask), ((uint32_t)rank << 18) | e);
    kept = __popcll(m);
    if (dbg != nullptr) {  // test-only branch census (gh_orb_plan_debug_counters)
      const uint64_t all_max = __ballot(ismax);
      if (lane == 0) {
        if (__popcll(cand) > kCap) atomicAdd(&dbg[kDbgRankDropped], 1u);
        if (strong_mask != 0ull && strong_mask != all_max) atomicAdd(&dbg[kDbgStrongSilenced], 1u);
        if (strong_mask == 0ull && all_max != 0ull) atomicAdd(&dbg[kDbgWeakCells], 1u);
      }
    }
  } else {
    int n = 0;
    bool strong = false;
    for (int base = 0; base < nz; base += 64) {
      const int i = base + lane;
      bool ismax = false;
      uint32_t e = 0;
      if (i < nz) {
        const uint32_t rc = list2[i];
        const uint8_t* sp = &score[(sy0 + (int)(rc >> 5)) * kScoreW + col0 + (int)(rc & 31u)];
        const int sv = sp[0];
        e = ((uint32_t)sv << 10) | rc;
        // branch-free: all 8 neighbour reads in flight at once, one comparison against their maximum
        const int n0 = sp[-kScoreW - 1], n1 = sp[-kScoreW], n2 = sp[-kScoreW + 1], n3 = sp[-1], n4 = sp[1],
                  n5 = sp[kScoreW - 1], n6 = sp[kScoreW], n7 = sp[kScoreW + 1];
        ismax = max3i(max3i(n0, n1, n2), max3i(n3, n4, n5), max(n6, n7)) < sv;
      }
      const uint64_t bm = __ballot(ismax);
      if (ismax) list[n + __popcll(bm & lt_mask)] = e ^ 0x3FFu;  // (the list holds KEYS: raster bits inverted, see the rank loop)
      n += __popcll(bm);
      strong = strong || (__ballot(ismax && (int)(e >> 10) > ini_th) != 0ull);
    }
    if (strong) {  // keep only candidates above the initial threshold (in place, order preserved)
      int m2 = 0;
      for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const uint32_t e = i < n ? list[i] : 0u;
        const bool keep = i < n && (int)(e >> 10) > ini_th;
        const uint64_t m = __ballot(keep);
        if (keep) list[m2 + __popcll(m & lt_mask)] = e;
        m2 += __popcll(m);
      }
      if (dbg != nullptr && lane == 0 && m2 != n) atomicAdd(&dbg[kDbgStrongSilenced], 1u);
      n = m2;
    }
    if (dbg != nullptr && lane == 0) {
      if (!strong && n > 0) atomicAdd(&dbg[kDbgWeakCells], 1u);
      atomicAdd(&dbg[kDbgDenseCells], 1u);
      if (n > kCap) atomicAdd(&dbg[kDbgRankDropped], 1u);
    }
    // rank within the cell by (score desc, raster asc) = by key, descending (list order is raster order and the raster position is
    // unique, so the key order is total): one comparison per pair; keep rank < cap, write in raster order
    for (int base = 0; base < n; base += 64) {
      const int i = base + lane;
      const uint32_t key = i < n ? list[i] : 0xFFFFFFFFu;
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += list[j] > key ? 1 : 0;
      const bool keep = i < n && rank < kCap;
      const uint64_t m = __ballot(keep);
      if (keep) put_entry(kept + __popcll(m & lt_mask), ((uint32_t)rank << 18) | (key ^ 0x3FFu));
      kept += __popcll(m);
    }
  }
  if (lane == 0) rec[0] = (uint32_t)kept;
  GH_PHASE(3);
  if (dbg != nullptr && lane == 0) {
    atomicAdd(&dbg[kDbgCells], 1u);
    if (kept > kCellRec - 1) atomicAdd(&dbg[kDbgOverflowCells], 1u);
    if (kept == kCap) atomicAdd(&dbg[kDbgCapCells], 1u);
    atomicMax(&dbg[kDbgMaxQueue], (uint32_t)nq);
    atomicMax(&dbg[kDbgMaxNz], (uint32_t)nz);
  }
}

// (SWAR variant: 8 waves per SIMD -- at most 64 VGPRs -- and 20.4 KB of LDS let 8 workgroups share a CU)
template <bool PK, int P1, bool PLANE = false>
__global__ __launch_bounds__(256, P1 != 0 ? GH_FAST_WAVES : 1) void fast_cells_kernel(LevelView lv, int ncx, int ncy, int min_th, int ini_th,
                                                         uint32_t* __restrict__ cell_cnt,
                                                         uint32_t* __restrict__ cell_ent, int cells_per_frame,
                                                         int cell_off, int n_frames, NextLevel nx,
                                                         uint32_t* __restrict__ dbg) {
  const int total = ((ncx + 1) >> 1) * ((ncy + 1) >> 1) * n_frames;
  const int tile_id = xcd_strip_tile(blockIdx.x, total);
  if (tile_id >= total) return;
  fast_cells_tile<PK, P1, PLANE>(lv, ncx, ncy, min_th, ini_th, cell_cnt, cell_ent, cells_per_frame, cell_off, n_frames, nx, dbg, tile_id);
}

// ------------------------------------------------------------------------------------------------
// EXPERIMENT, round 6: the score plane by WAVE-AUTONOMOUS SLIDING WINDOWS -- no block-wide barrier, no shared tile.
// fast_cells_tile stages a 64 x 64 tile for 256 threads and crosses three __syncthreads; its tile stages reach ~60 % of the issue
// rate between them (docs/notes_r05.md).  Here a WAVE owns a strip of 256 pixel columns (one dword per lane; 248 of them are its
// own, the rest is the +-3 px halo) and marches down kSwRows rows of one frame:
//   * the image rows it needs live in a wave-private LDS ring of 16 rows x 256 B, fed from registers that were loaded two row
//     groups (8 rows) ahead -- the HBM latency is covered by the wave's own work, not by occupancy;
//   * pass 1 (the SWAR compass test of fast_cells_tile, same arithmetic) runs on 4 rows at a time, its survivors are appended to a
//     wave-private queue with ONE wave scan per group;
//   * pass 2 (exact arc score) pops 64 survivors at a time, so every lane is busy whatever the candidate density, reads its 17
//     pixels from the ring and drops the score byte into a wave-private score ring (8 rows);
//   * a row of the score ring leaves as 62 aligned dwords (one 248-byte run of the plane) once the survivors of its group are
//     done -- at most one group later -- and is cleared for the row eight below.
// Only wave-level ordering is needed (LDS operations of a wave execute in order; the fences below are compiler fences).
// Output = the plane fast_cells_kernel<.., PLANE> writes (S of oracle step 2 / 3; pixel (y, x) at plane[y * pitch + x + kQtPlaneX]),
// rows kEdge .. h - 1, bit for bit.  GSLAM_HIP_ORB_PLANE_SW=1 selects it for the quadtree mode (the pyramid then comes from the
// stand-alone resize launches).  Result of the experiment: profiles/orb_sliding_window_r06.txt, DESIGN.md 6a.
constexpr int kSwOwn = 248;    // pixels a strip owns: columns xs + 3 .. xs + 250 of its 256 (plane dwords are aligned at x = 3 mod 4)
constexpr int kSwRows = 64;    // rows per wave (128: 1.49 ms per 100 x 1080p against 1.38 -- fewer, longer waves fill the chip worse)
constexpr int kSwGroup = 4;    // rows per pass-1 group (their candidate flags share one register: bits 15 - 2 t, 14 - 2 t, 31 - 2 t, 30 - 2 t)
constexpr int kSwRing = 16;    // image ring: rows y - 3 .. y + 6 of the current group + the 4 rows of the group before it (their leftover survivors)
constexpr int kSwSRing = 8;    // score ring: two groups
constexpr int kSwQueue = 4 * kSwOwn + 64 + 32;  // a whole group of candidates + the leftover of the one before
struct SwLds {
  uint32_t img[kSwRing * 64];
  uint32_t sc[kSwSRing * 64];
  uint16_t q[kSwQueue];
};
__device__ __forceinline__ void sw_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(256) void fast_plane_sw_kernel(LevelView lv, int min_th, int nstrips, int nchunks, int n_frames,
                                                            uint8_t* __restrict__ plane, size_t plane_frame_stride, int plane_pitch,
                                                            int variant) {
  __shared__ __attribute__((aligned(16))) SwLds lds_all[4];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  SwLds& L = lds_all[wv];
  const int total = nstrips * nchunks * n_frames;
  const int wid = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wv);
  if (wid >= total) return;
  const int frame = wid / (nstrips * nchunks), rem = wid - frame * (nstrips * nchunks);
  const int chunk = rem / nstrips, strip = rem - chunk * nstrips;
  const int xs = kSwOwn * strip;                                        // first image column of the strip (a multiple of 4)
  const int y_begin = kEdge + kSwRows * chunk;
  const int y_out_end = min(y_begin + kSwRows, lv.h);                   // rows written (zeros below the valid region)
  const int y_valid_end = min(y_out_end, lv.h - kEdge);                 // rows scored
  const int ngroups = (y_out_end - y_begin + kSwGroup - 1) / kSwGroup;
  const uint8_t* img = lv.base + (size_t)frame * lv.frame_stride;
  uint8_t* pl = plane + (size_t)frame * plane_frame_stride;
  const uint32_t gx = (uint32_t)min(xs + 4 * lane, lv.pitch - 4);
  auto load_row = [&](int y) -> uint32_t {
    const int yc = y < 0 ? 0 : (y > lv.h - 1 ? lv.h - 1 : y);
    return *reinterpret_cast<const uint32_t*>(img + (__umul24((uint32_t)yc, (uint32_t)lv.pitch) + gx));
  };
  // The PREFETCHED rows are loaded by inline asm and waited for by hand.  The compiler's s_waitcnt insertion treats a counter with
  // loads AND stores pending as out of order and drains it (vmcnt(0)) at the top of every other group -- the plane stores of the
  // flush are always pending -- which cut the prefetch distance from two groups to none.  Loads return in order among themselves,
  // so "at most 4 operations outstanding" implies that everything older than the 4 youngest loads has landed, whatever the stores
  // do (a pending store can only make the wait longer).  The registers must not be copied between the two asm statements (the
  // compiler believes they are valid at once): 48 of 128 registers are in use, and tests/test_build_isa.py looks at the code.
  auto load_row_async = [&](int y, uint32_t& dst) {
    const int yc = y < 0 ? 0 : (y > lv.h - 1 ? lv.h - 1 : y);
    const uint32_t off = __umul24((uint32_t)yc, (uint32_t)lv.pitch) + gx;
    asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(off), "s"(img) : "memory");
  };
  auto ring_row = [&](int y) { return (uint32_t)(y - y_begin + 3) & (kSwRing - 1); };
  // rows y_begin - 3 .. y_begin + 2 now, the new rows of groups 0 and 1 in flight
  {
    uint32_t v[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) v[r] = load_row(y_begin - 3 + r);
#pragma unroll
    for (int r = 0; r < 6; ++r) L.img[r * 64 + lane] = v[r];
  }
  uint32_t pfA[kSwGroup], pfB[kSwGroup];
#pragma unroll
  for (int r = 0; r < kSwGroup; ++r) load_row_async(y_begin + 3 + r, pfA[r]);
#pragma unroll
  for (int r = 0; r < kSwGroup; ++r) load_row_async(y_begin + 3 + kSwGroup + r, pfB[r]);
#pragma unroll
  for (int r = 0; r < kSwSRing; ++r) L.sc[r * 64 + lane] = 0u;
  // which of its four pixels a lane owns: lane 0 only pixel 3, lane 62 pixels 0 .. 2, lane 63 none
  const uint32_t own = lane == 0 ? 0x55000000u : (lane == 62 ? 0xAA00FF00u : (lane == 63 ? 0u : 0xFF00FF00u));
  constexpr uint32_t kF = 0x00FF00FFu;
  const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane(min(max(min_th, 0), 255));
  const uint32_t kAe = (0x8000u + t) * 0x10001u, kDe = (0x8000u - t - 1u) * 0x10001u;
  const uint32_t kAo = (0x4000u + t) * 0x10001u, kDo = (0x4000u - t - 1u) * 0x10001u;
  const int lm = lane > 0 ? lane - 1 : 0, lp = lane < 63 ? lane + 1 : 63;
  int q_n = 0, q_old = 0;  // entries in the queue; how many of them are left over from the group before (wave-uniform)
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  (void)lt_mask;

  // pass 2 on the queue entries [head, head + cnt), cnt <= 64
  auto pass2 = [&](int head, int cnt) {
    if (lane < cnt) {
      const uint32_t e = L.q[head + lane];
      const uint32_t rr = e >> 8, col = e & 255u;
      const uint8_t* ib = reinterpret_cast<const uint8_t*>(L.img);
      auto px = [&](int dy, int dx) -> int { return ib[((rr + (uint32_t)(dy + kSwRing)) & (kSwRing - 1)) * 256u + col + dx]; };
      const int c = px(0, 0);
      int r[16];
      r[0] = px(-3, 0);  r[1] = px(-3, 1);  r[2] = px(-2, 2);   r[3] = px(-1, 3);
      r[4] = px(0, 3);   r[5] = px(1, 3);   r[6] = px(2, 2);    r[7] = px(3, 1);
      r[8] = px(3, 0);   r[9] = px(3, -1);  r[10] = px(2, -2);  r[11] = px(1, -3);
      r[12] = px(0, -3); r[13] = px(-1, -3); r[14] = px(-2, -2); r[15] = px(-3, -1);
      const int s = fast_score16_pk(r, c);
      const int x = xs + (int)col;
      if (s > min_th && x >= kEdge && x < lv.w - kEdge)
        reinterpret_cast<uint8_t*>(L.sc)[((rr - 3u) & (kSwSRing - 1)) * 256u + col - 3u] = (uint8_t)s;
    }
  };
  // rows [y0, y0 + 4) of the score ring to the plane, then cleared
  auto flush = [&](int y0) {
#pragma unroll
    for (int r = 0; r < kSwGroup; ++r) {
      const int y = y0 + r;
      const uint32_t idx = ((uint32_t)(y - y_begin) & (kSwSRing - 1)) * 64u + (uint32_t)lane;
      const uint32_t v = L.sc[idx];
      L.sc[idx] = 0u;
      // UNCONDITIONAL store: a lane without a dword of its own writes to the row's left margin (plane columns 0 .. 3: pixel x sits
      // at column x + kQtPlaneX, nobody reads the margin).  A branch around the store costs the compiler its count of the memory
      // operations in flight: it then drains the prefetched rows (s_waitcnt vmcnt(0)) at the top of every other group.
      const int pc = xs + 3 + kQtPlaneX + 4 * lane;  // plane column of the lane's first owned pixel (a multiple of 4)
      const bool mine = lane < 62 && y < y_out_end && xs + 3 + 4 * lane < lv.w && pc + 3 < plane_pitch;
      const int yy = y < lv.h ? y : lv.h - 1;
      if (variant & 2) {  // (A/B: the conditional store of the first version)
        if (mine) *reinterpret_cast<uint32_t*>(pl + (__umul24((uint32_t)yy, (uint32_t)plane_pitch) + (uint32_t)pc)) = v;
      } else {
        *reinterpret_cast<uint32_t*>(pl + (__umul24((uint32_t)yy, (uint32_t)plane_pitch) + (uint32_t)(mine ? pc : 0))) = v;
      }
    }
  };
  auto group = [&](int g, uint32_t (&pf)[kSwGroup]) {
    const int y = y_begin + kSwGroup * g;
    // the group's new rows y + 3 .. y + 6 (asked for two groups ago; the 4 loads of the group in between may still be in
    // flight) into the ring; their registers go back out for the rows two groups on
    static_assert(kSwGroup == 4, "the wait below names four registers and leaves four loads in flight");
    asm volatile("s_waitcnt vmcnt(4)" : "+v"(pf[0]), "+v"(pf[1]), "+v"(pf[2]), "+v"(pf[3])::"memory");
#pragma unroll
    for (int r = 0; r < kSwGroup; ++r) L.img[ring_row(y + 3 + r) * 64u + (uint32_t)lane] = pf[r];
    if (variant & 8) {  // (A/B: compiler-tracked loads)
#pragma unroll
      for (int r = 0; r < kSwGroup; ++r) pf[r] = load_row(y + 3 + 2 * kSwGroup + r);
    } else {
#pragma unroll
      for (int r = 0; r < kSwGroup; ++r) load_row_async(y + 3 + 2 * kSwGroup + r, pf[r]);
    }
    sw_fence();
    // ---- pass 1: SWAR compass test (fast_cells_tile, P1 = 1) on the rows y .. y + 3
    uint32_t allbits = 0;
#pragma unroll
    for (int tr = 0; tr < kSwGroup; ++tr) {
      const uint32_t rc = ring_row(y + tr) * 64u, ru = ring_row(y + tr - 3) * 64u, rd = ring_row(y + tr + 3) * 64u;
      const uint32_t wc = L.img[rc + lane], wl = L.img[rc + lm], wr = L.img[rc + lp], wu = L.img[ru + lane], wd = L.img[rd + lane];
      constexpr uint32_t kOdd = 0x0c030c01u;
      const uint2 C{wc & kF, __builtin_amdgcn_perm(0u, wc, kOdd)}, U{wu & kF, __builtin_amdgcn_perm(0u, wu, kOdd)},
          D{wd & kF, __builtin_amdgcn_perm(0u, wd, kOdd)};
      const uint32_t le = __builtin_amdgcn_perm(0u, wl, kOdd), re = __builtin_amdgcn_perm(wr, wc, 0x0c050c03u);
      const uint32_t lo = __builtin_amdgcn_perm(wc, wl, 0x0c040c02u), ro = wr & kF;
      auto half = [](uint32_t A, uint32_t Dk, uint32_t u, uint32_t d, uint32_t l, uint32_t r) {
        const uint32_t not_bright_lr = (A - l) & (A - r);
        const uint32_t bright = __builtin_amdgcn_bitop3_b32(A - u, A - d, not_bright_lr, 0x15);  // ~(a & b) & ~c
        const uint32_t dark_lr = (Dk - l) | (Dk - r);
        const uint32_t dark = __builtin_amdgcn_bitop3_b32(Dk - u, Dk - d, dark_lr, 0xA8);  // (a | b) & c
        return bright | dark;
      };
      const uint32_t ye = half(C.x + kAe, C.x + kDe, U.x, D.x, le, re);
      const uint32_t yo = half(C.y + kAo, C.y + kDo, U.y, D.y, lo, ro);
      const uint32_t w = __builtin_amdgcn_bitop3_b32(ye, yo, 0x80008000u, 0xE4);  // (a & c) | (b & ~c)
      const uint32_t rowmask = y + tr < y_valid_end ? 0xC000C000u >> (2 * tr) : 0u;
      allbits = __builtin_amdgcn_bitop3_b32(allbits, w >> (2 * tr), rowmask, 0xF8);  // a | (b & c)
    }
    allbits &= own;
    // ---- the survivors join the queue: one wave scan per group
    {
      const int cnt = __popc(allbits);
      const int incl = wave_incl_scan_i32(cnt);
      const int wave_total = __builtin_amdgcn_readlane(incl, 63);
      int base = q_n + incl - cnt;
      const uint32_t rr0 = ring_row(y);
      while (allbits) {
        const int b = __ffs((int)allbits) - 1;
        const uint32_t bb = 15u - ((uint32_t)b & 15u), tr = bb >> 1, k = 2u * ((uint32_t)b >> 4) + (bb & 1u);
        L.q[base++] = (uint16_t)((((rr0 + tr) & (kSwRing - 1)) << 8) | (4u * (uint32_t)lane + k));
        allbits &= allbits - 1u;
      }
      q_n += wave_total;
    }
    sw_fence();
    // ---- pass 2: full chunks, then whatever is still left of the group before (its rows leave the ring next group)
    // (the leftover of the group before first, then ITS rows go out -- half a group ahead of the next wait for prefetched rows,
    //  which would otherwise sit behind plane stores issued a moment ago -- then the rest of the full chunks)
    int head = 0;
    while (head < q_old) {
      const int cnt = min(64, q_n - head);
      pass2(head, cnt);
      head += cnt;
    }
    sw_fence();
    if (g > 0 && !(variant & 4)) flush(y - kSwGroup);
    while (q_n - head >= 64) {
      pass2(head, 64);
      head += 64;
    }
    sw_fence();
    // the leftover (< 64 entries, all of this group) to the front
    const int left = q_n - head;
    if (head > 0 && left > 0) {
      const uint16_t e = lane < left ? L.q[head + lane] : (uint16_t)0;
      sw_fence();
      if (lane < left) L.q[lane] = e;
    }
    q_n = q_old = left;
    sw_fence();
    if (g > 0 && (variant & 4)) flush(y - kSwGroup);  // (A/B: the first version's place)
  };
  for (int g = 0; g < ngroups; g += 2) {
    group(g, pfA);
    if (g + 1 < ngroups) group(g + 1, pfB);
  }
  // the last group's leftover, then its rows
  sw_fence();
  if (q_n > 0) pass2(0, q_n);
  sw_fence();
  flush(y_begin + kSwGroup * (ngroups - 1));
}

// Every level in ONE launch, over a pyramid that exists already (stand-alone resize launches): what a small call wants --
// with the next level fused into fast_cells(l) the eight levels are eight DEPENDENT launches of ~14 us of latency each,
// whatever the image size; here the dependent chain is seven small resize launches and one FAST launch.
struct AllLevels {
  LevelView lv[kMaxL];
  int ncx[kMaxL], ncy[kMaxL], cell_off[kMaxL], tile_start[kMaxL + 1];
  int n_levels;
};
template <bool PK, int P1>
__global__ __launch_bounds__(256) void fast_cells_all_kernel(AllLevels A, int min_th, int ini_th,
                                                             uint32_t* __restrict__ cell_cnt,
                                                             uint32_t* __restrict__ cell_ent, int cells_per_frame,
                                                             int n_frames, uint32_t* __restrict__ dbg) {
  int l = 0;
  for (int k = 1; k < A.n_levels; ++k)
    if ((int)blockIdx.x >= A.tile_start[k]) l = k;
  const NextLevel none{nullptr, 0, 0, 0, ResizeTabs{nullptr, nullptr, nullptr, nullptr}, nullptr, nullptr, -1};
  fast_cells_tile<PK, P1>(A.lv[l], A.ncx[l], A.ncy[l], min_th, ini_th, cell_cnt, cell_ent, cells_per_frame, A.cell_off[l], n_frames, none,
                      dbg, (int)blockIdx.x - A.tile_start[l]);
}

// ------------------------------------------------------------------------------------------------
// block-wide exclusive scan of one int per thread (256 threads); returns exclusive prefix, *total = sum
__device__ __forceinline__ int block_excl_scan(int v, int* wave_tot /* shared[5] */, int* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  __syncthreads();  // protect wave_tot reuse
  if (lane == 63) wave_tot[wv] = incl;
  __syncthreads();
  int off = 0;
  for (int k = 0; k < wv; ++k) off += wave_tot[k];
  *total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
  return off + incl - v;
}

// Entries 7 .. n-1 of a cell, from its 128-byte slot: entry 7 alone, then four per 16-byte load -- a dense cell costs
// 1 + (n - 8) / 4 dependent memory round trips per pass instead of n - 7 (the slowest thread of a workgroup sets its pace).
template <typename F>
__device__ __forceinline__ void for_each_overflow(const uint32_t* __restrict__ ovf, int n, F f) {
  static_assert(kCellRec - 1 == 7 && kCap % 4 == 0, "entry 8 starts the 16-byte groups");
  if (n > 7) f(ovf[7]);
  for (int e = 8; e < n; e += 4) {
    const uint4 w = *reinterpret_cast<const uint4*>(ovf + e);
    f(w.x);
    if (e + 1 < n) f(w.y);
    if (e + 2 < n) f(w.z);
    if (e + 3 < n) f(w.w);
  }
}

// Visit the entries of one cell.  rec = the cell's 32-byte record {count, entries 0..6} (two 16-byte loads issued
// together: one memory round trip for the common case), ovf = its 128-byte slot holding entries 7.. at their index.
template <typename F>
__device__ __forceinline__ int for_each_entry(const uint32_t* __restrict__ rec, const uint32_t* __restrict__ ovf, F f) {
  const uint4 a = reinterpret_cast<const uint4*>(rec)[0];
  const uint4 b = reinterpret_cast<const uint4*>(rec)[1];
  const int n = (int)a.x;
  const uint32_t first[kCellRec - 1] = {a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int e = 0; e < kCellRec - 1; ++e)
    if (e < n) f(first[e]);
  for_each_overflow(ovf, n, f);
  return n;
}

// The same on a record that is already in registers
template <typename F>
__device__ __forceinline__ int for_each_entry_reg(const uint4& a, const uint4& b, const uint32_t* __restrict__ ovf, F f) {
  const int n = (int)a.x;
  const uint32_t first[kCellRec - 1] = {a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int e = 0; e < kCellRec - 1; ++e)
    if (e < n) f(first[e]);
  for_each_overflow(ovf, n, f);
  return n;
}

struct SelectArgs {
  int ncells[kMaxL], cell_off[kMaxL], ncx[kMaxL], quota[kMaxL], quota_off[kMaxL];
};

// CACHED (levels of at most kSelCached * 256 cells): every thread fetches the records of ALL its cells up front -- one
// memory round trip with up to kSelCached loads in flight -- and the three passes below (histogram, counts, output) read
// them from registers.  Streaming them three times made the kernel a chain of ~24 dependent round trips per thread at
// 1080p (the workgroup does little else: it was latency, not bandwidth).
constexpr int kSelCached = 8;
template <bool CACHED>
__global__ __launch_bounds__(256) void select_kernel(SelectArgs a, const uint32_t* __restrict__ cell_cnt,
                                                     const uint32_t* __restrict__ cell_ent, int cells_per_frame,
                                                     int K, SelKp* __restrict__ sel, int32_t* __restrict__ level_cnt,
                                                     uint32_t* __restrict__ dbg, int level0) {
  // bin k lives at k + (k >> 5): a thread's 32 consecutive bins (tid * 32 + i) then fall into different banks for
  // different threads (unpadded, all 64 lanes of a wave hit bank i)
  __shared__ uint32_t hist[kHistBins + kHistBins / 32];
  __shared__ int wave_tot[4];
  __shared__ int s_cut, s_m;
  const int l = level0 + blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int ncells = a.ncells[l], quota = a.quota[l];
  const uint32_t* rec = cell_cnt + ((size_t)b * cells_per_frame + a.cell_off[l]) * kCellRec;
  const uint32_t* ent = cell_ent + ((size_t)b * cells_per_frame + a.cell_off[l]) * kCap;
  if (quota <= 0 || ncells <= 0) {
    if (tid == 0) level_cnt[b * kMaxL + l] = 0;
    return;
  }
  uint4 ra[kSelCached], rb[kSelCached];
  if (CACHED) {
#pragma unroll
    for (int ci = 0; ci < kSelCached; ++ci) {
      const int c = ci * 256 + tid;
      ra[ci] = make_uint4(0u, 0u, 0u, 0u);
      rb[ci] = make_uint4(0u, 0u, 0u, 0u);
      if (c < ncells) {
        ra[ci] = reinterpret_cast<const uint4*>(rec + (size_t)c * kCellRec)[0];
        rb[ci] = reinterpret_cast<const uint4*>(rec + (size_t)c * kCellRec)[1];
      }
    }
  }
  for (int i = tid; i < kHistBins + kHistBins / 32; i += 256) hist[i] = 0;
  __syncthreads();
  auto hist_add = [&](uint32_t v) {
    const int ck = (int)(v >> 18) * 256 + (255 - (int)((v >> 10) & 255));
    atomicAdd(&hist[ck + (ck >> 5)], 1u);
  };
  if (CACHED) {
#pragma unroll
    for (int ci = 0; ci < kSelCached; ++ci) {  // compile-time indices: the records stay in registers
      const int c = ci * 256 + tid;
      if (c < ncells) for_each_entry_reg(ra[ci], rb[ci], ent + (size_t)c * kCap, hist_add);
    }
  } else {
    for (int c = tid; c < ncells; c += 256) for_each_entry(rec + (size_t)c * kCellRec, ent + (size_t)c * kCap, hist_add);
  }
  __syncthreads();
  // cut-off bin: smallest bin whose inclusive prefix reaches the quota
  constexpr int kPer = kHistBins / 256;
  int mine = 0;
  static_assert(kPer == 32, "the padding below assumes 32 bins per thread");
  for (int i = 0; i < kPer; ++i) mine += (int)hist[tid * (kPer + 1) + i];
  int total;
  const int excl = block_excl_scan(mine, wave_tot, &total);
  if (tid == 0) {
    s_cut = kHistBins;  // everything selected
    s_m = 0;
  }
  __syncthreads();
  if (total > quota && excl < quota && excl + mine >= quota) {
    int run = excl;
    for (int i = 0; i < kPer; ++i) {
      const int hv = (int)hist[tid * (kPer + 1) + i];
      if (run + hv >= quota) {
        s_cut = tid * kPer + i;
        s_m = quota - run;
        if (dbg != nullptr) {
          atomicAdd(&dbg[kDbgSelCut], 1u);
          if (quota - run < hv) atomicAdd(&dbg[kDbgSelTieSplit], 1u);
        }
        break;
      }
      run += hv;
    }
  }
  __syncthreads();
  const int cut = s_cut, m = s_m;
  // slots: traverse cells in order, chunks of 256 with carries
  int tie_carry = 0, out_carry = 0;
  SelKp* out = sel + (size_t)b * K + a.quota_off[l];
  const int ncx = a.ncx[l];
  // one chunk of 256 cells: cell c of this thread, its entries visited through `each(f)`
  auto chunk = [&](int c, auto each) {
    int n_lt = 0, n_eq = 0;
    if (c < ncells) {
      each([&](uint32_t v) {
        const int ck = (int)(v >> 18) * 256 + (255 - (int)((v >> 10) & 255));
        n_lt += ck < cut;
        n_eq += ck == cut;
      });
    }
    int tot_eq, tot_sel;
    const int tie_excl = tie_carry + block_excl_scan(n_eq, wave_tot, &tot_eq);
    int taken = m - tie_excl;
    taken = taken < 0 ? 0 : (taken > n_eq ? n_eq : taken);
    const int mysel = n_lt + taken;
    const int out_excl = out_carry + block_excl_scan(mysel, wave_tot, &tot_sel);
    if (mysel > 0) {
      const int cx = c % ncx, cy = c / ncx;
      int k = 0, ties = 0;
      if (dbg != nullptr && n_lt + n_eq > kCellRec - 1) atomicAdd(&dbg[kDbgSelOverflowCells], 1u);
      each([&](uint32_t v) {
        const int s = (int)((v >> 10) & 255);
        const int ck = (int)(v >> 18) * 256 + (255 - s);
        bool take = ck < cut;
        if (ck == cut && ties < taken) {
          take = true;
          ++ties;
        }
        if (take) {
          SelKp o;
          o.x = (uint16_t)(kEdge + cx * kCell + (int)(v & 31));
          o.y = (uint16_t)(kEdge + cy * kCell + (int)((v >> 5) & 31));
          o.score = (uint8_t)s;
          o.level = (uint8_t)l;
          o.pad = 0;
          out[out_excl + k] = o;
          ++k;
        }
      });
    }
    tie_carry += tot_eq;
    out_carry += tot_sel;
  };
  if (CACHED) {
#pragma unroll
    for (int ci = 0; ci < kSelCached; ++ci) {
      if (ci * 256 < ncells) {  // (uniform: the block scans inside are reached by every thread or by none)
        const int c = ci * 256 + tid;
        chunk(c, [&](auto f) { for_each_entry_reg(ra[ci], rb[ci], ent + (size_t)c * kCap, f); });
      }
    }
  } else {
    for (int cb = 0; cb < ncells; cb += 256) {
      const int c = cb + tid;
      chunk(c, [&](auto f) { for_each_entry(rec + (size_t)c * kCellRec, ent + (size_t)c * kCap, f); });
    }
  }
  if (tid == 0) level_cnt[b * kMaxL + l] = out_carry;
  if (dbg != nullptr && tid == 0 && out_carry < quota) atomicAdd(&dbg[kDbgStarvedLevels], 1u);
}

// ------------------------------------------------------------------------------------------------
struct DescribeArgs {
  LevelView lv[kMaxL];
  int quota[kMaxL], quota_off[kMaxL];
  float scale[kMaxL];
  int nlevels;
  // describe_pipe_kernel: workgroups are dealt PER LEVEL (blk_off[l] = first workgroup of level l within a frame, blk_off[nlevels] =
  // workgroups per frame; a workgroup = 4 kDescPipe slots of ONE level's quota)
  int blk_off[kMaxL + 1];
};

// Sum of one int per lane over the wave, returned in every lane: four DPP adds give each 16-lane row its row sum
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), the four row sums meet through v_readlane / SALU adds.
// (A __shfl_xor butterfly is 6 x 2 ds_bpermute plus the index arithmetic.)
__device__ __forceinline__ int wave_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);  // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);  // row_mirror
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
         __builtin_amdgcn_readlane(v, 48);
}

// Patch geometry of orb_describe for a test pattern of radius BR (blurred region = (2 BR + 1)^2, raw patch 3 px wider for
// the 7x7 Gaussian): BR = 13 is the 30-bin table mode (33 x 33 patch: also covers the radius-15 centroid disc), BR = 19 the
// continuous-steering mode, whose rotated points reach radius 19 (45 x 45 patch).
template <int BR>
struct DescGeom {
  static constexpr int kC = BR + 3;                         // patch radius
  static constexpr int kPatch = 2 * kC + 1;                  // 33 / 45
  static constexpr int kPatchPitch = (kPatch + 6) & ~3;      // 36 / 48: whole dwords that cover any kPatch-byte run
  static constexpr int kRowDw = kPatchPitch / 4;
  static constexpr int kBlur = 2 * BR + 1;                   // 27 / 39
  static constexpr int kBlurPitch = (kBlur + 3) & ~3;        // 28 / 40
  static constexpr int kGroups = kBlurPitch / 4;             // 4-column groups per blur row
};
constexpr int kBlurPitch = DescGeom<13>::kBlurPitch;  // (the table mode's pitch: upload_pattern bakes it into the offsets)

// Continuous steering (gh_orb_plan_set_steering, oracle/orb_oracle.c steps 6' and 8'): the orientation is the fp32
// polynomial arctangent OpenCV's fastAtan2 uses (ORB-SLAM's IC_Angle calls it), and every test point is rotated by it.
// Plain fp32 + - * / in a fixed order, no contraction, so that the CPU checker reproduces every bit.
__host__ __device__ inline float orb_fast_atan2_deg(float y, float x) {
  const float p1 = 57.283627f, p3 = -18.667446f, p5 = 8.9140005f, p7 = -2.5397246f;  // 0.99978784, -0.32580840, 0.15557865, -0.044326555 x (float)(180 / pi)
  const float eps = 2.2204460492503131e-16f;
  const float ax = fabsf(x), ay = fabsf(y);
  float a;
  if (ax >= ay) {
    const float c = ay / (ax + eps), c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    const float c = ax / (ay + eps), c2 = c * c;
    a = 90.0f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0.0f) a = 180.0f - a;
  if (y < 0.0f) a = 360.0f - a;
  return a;
}
// cos / sin of an angle in degrees, [0, 360]: nearest quadrant k, remainder r in [-45, 45] degrees, Taylor polynomials
__host__ __device__ inline void orb_sincos_deg(float a, float* cs, float* sn) {
  const int k = (int)(a / 90.0f + 0.5f);
  const float r = (a - 90.0f * (float)k) * 0.017453292f, r2 = r * r;
  const float s = r * (1.0f + r2 * (-0.16666667f + r2 * (0.0083333338f + r2 * -0.00019841270f)));
  const float c = 1.0f + r2 * (-0.5f + r2 * (0.041666668f + r2 * (-0.0013888889f + r2 * 0.000024801588f)));
  switch (k & 3) {
    case 0: *cs = c; *sn = s; break;
    case 1: *cs = -s; *sn = c; break;
    case 2: *cs = -c; *sn = -s; break;
    default: *cs = s; *sn = -c; break;
  }
}
__host__ __device__ inline int orb_reflect101(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// The h-pass of the 7x7 blur as a banded matrix product on the (otherwise idle) matrix cores -- describe_kernel<13, false, true>.
// B operand of v_mfma_f32_16x16x32_f16 for the horizontal taps: B[k][n] = g[k - n] (0 <= k - n <= 6), the same for both
// 16-column blocks because each block has its own K window (patch columns 16 nb .. 16 nb + 31).  Lane l supplies column
// n = l & 15, k = 8 (l >> 4) + e, e = 0..7 as eight f16 (the taps 144 / 268 / 391 / 442 are exact in f16).
struct BlurBTable { uint32_t w[64][4]; };
constexpr uint32_t f16_bits_of_int(int v) {  // 0 <= v < 2048: exact
  if (v == 0) return 0u;
  int e = 0;
  while ((v >> (e + 1)) != 0) ++e;
  const int mant = e <= 10 ? (v << (10 - e)) : (v >> (e - 10));  // (2048: e = 11 -- no shift by a negative count)
        return (uint32_t)(((e + 15) << 10) | (mant & 0x3FF));
}
constexpr BlurBTable make_blur_b() {
  BlurBTable t{};
  const int g[7] = {144, 268, 391, 442, 391, 268, 144};
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < 8; ++e) {
      const int d = 8 * (l >> 4) + e - (l & 15);
      const uint32_t h = (d >= 0 && d <= 6) ? f16_bits_of_int(g[d]) : 0u;
      t.w[l][e >> 1] |= h << (16 * (e & 1));
    }
  return t;
}
__device__ const BlurBTable kBlurB = make_blur_b();

// Layout of the blurred patch the MFMA variant leaves in LDS: column-major, one 32-byte line per blur column, row r at byte
// 8 (r / 7) + r % 7 of the line (each lane of the v-pass owns 7 consecutive rows of one column and stores them as one
// 8-byte write).  upload_pattern bakes either layout into the per-bin offset table.
__host__ __device__ constexpr int blur_offset_mfma(int row, int col) { return col * 32 + 8 * (row / 7) + row % 7; }

// Intensity-centroid moments of a patch in LDS (oracle step 6): rows of kRowDw dwords, patch column 0 at byte xoff of a row,
// patch centre at [kC][kC].  Returns the wave sums in every lane.
template <int kC, int kRowDw>
__device__ __forceinline__ void desc_moments(const uint8_t* patch, uint32_t xoff, int lane, int* m10_out, int* m01_out) {
  // intensity centroid over the radius-15 disc (patch centre at [kC][kC]).  Lane = (disc row, half): 16 bytes of
  // the row as 4 dwords (unaligned start: 5 dword reads + v_alignbyte), bytes outside |u| <= u_max(|v|) masked off,
  // then two v_dot4_u32_u8 per dword: sum I and sum (u + 16) I  ->  m10 = sum (u + 16) I - 16 sum I, m01 = v sum I.
  int m10 = 0, m01 = 0;
  if (lane < 62) {
    const int r = lane >> 1, h = lane & 1;
    const int v = r - 15, av = v < 0 ? -v : v;
    const int umax = (int)((0x3689ABCDDEEEFFFFull >> (4 * av)) & 15ull);  // GH_ORB_UMAX as nibbles
    const uint32_t boff = xoff + (uint32_t)(kC - 15) + 16u * (uint32_t)h;  // byte offset of u = -15 + 16 h in patch row r + kC - 15
    const uint32_t* q = reinterpret_cast<const uint32_t*>(patch) + (r + kC - 15) * kRowDw + (boff >> 2);
    const uint32_t sh = boff & 3u;
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
    const uint32_t w[4] = {__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh),
                           __builtin_amdgcn_alignbyte(d3, d2, sh), __builtin_amdgcn_alignbyte(d4, d3, sh)};
    // kept byte positions p (u = -15 + 16 h + p): h = 0: p >= 15 - umax;  h = 1: p <= umax - 1
    const int lo = h ? 0 : 15 - umax, hi = h ? umax - 1 : 15;
    uint32_t sI = 0, sW = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nlo = min(max(lo - 4 * j, 0), 4), nhi = min(max(4 * j + 3 - hi, 0), 4);
      const uint32_t mask = (uint32_t)(0xFFFFFFFFull << (8 * nlo)) & (uint32_t)(0xFFFFFFFFull >> (8 * nhi));
      const uint32_t I4 = w[j] & mask;
      const uint32_t wgt = 0x04030201u + 0x04040404u * (uint32_t)j + (h ? 0x10101010u : 0u);  // u + 16 per byte
      sI = __builtin_amdgcn_udot4(I4, 0x01010101u, sI, false);
      sW = __builtin_amdgcn_udot4(I4, wgt, sW, false);
    }
    m10 = (int)sW - 16 * (int)sI;
    m01 = __mul24(v, (int)sI);  // |v| <= 15, sI < 2^13
  }
  *m10_out = wave_sum_i32(m10);
  *m01_out = wave_sum_i32(m01);
}

// Orientation bin of the 30-bin table mode from the moments: the first bin whose direction has the centroid to its right
// while the previous one has it to its left (integer cross products, oracle step 6).
__device__ __forceinline__ int desc_bin(int dir_x, int dir_y, int m10, int m01, int lane);
__device__ __forceinline__ int desc_bin(const int32_t* __restrict__ dir, int m10, int m01, int lane) {
  return desc_bin(lane < GH_ORB_NBINS ? dir[2 * lane] : 0, lane < GH_ORB_NBINS ? dir[2 * lane + 1] : 0, m10, m01, lane);
}
// (dir_x, dir_y) = this lane's direction (lanes >= GH_ORB_NBINS: anything)
__device__ __forceinline__ int desc_bin(int dir_x, int dir_y, int m10, int m01, int lane) {
  long long c = 0;
  if (lane < GH_ORB_NBINS) c = (long long)dir_x * m01 - (long long)dir_y * m10;
  const int c_neg = c < 0 ? 1 : 0;
  const int prev_lane = (lane + GH_ORB_NBINS - 1) % GH_ORB_NBINS;
  const int prev_neg = __shfl(c_neg, prev_lane);
  const uint64_t hit = __ballot(lane < GH_ORB_NBINS && !prev_neg && c_neg);
  int bin = 0;
  if ((m10 != 0 || m01 != 0) && hit != 0ull) bin = __ffsll((unsigned long long)hit) - 1;
  return bin;
}

// The 7x7 blur of a 33 x 33 patch (+ 4 scratch rows) in LDS, rows of 9 dwords with patch column 0 at byte 0, into the
// column-major 1 KB blurred patch bl (blur_offset_mfma) -- orb_describe's MFMA formulation.
__device__ __forceinline__ uint4 desc_blur_b(int lane) { return *reinterpret_cast<const uint4*>(kBlurB.w[lane]); }
__device__ __forceinline__ void desc_blur_mfma(const uint8_t* patch, uint8_t* bl, int lane, const uint4 bw) {
  constexpr int kRowDw = 9;
  // Separable 7x7 integer Gaussian, horizontal pass on the matrix cores, vertical pass out of the accumulators.
  //   h-pass   D[m][n] = sum_k A[m][k] B[k][n]:  A[m][k] = f16(1024 + P[row(m)][16 nb + k]) (a byte OR 0x6400 is that f16,
  //            exact), B = the banded tap matrix kBlurB, C = 0.  The byte bias adds 1024 * 2048: D = 2^21 + H with H < 2^19,
  //            a float in [2^21, 2^22) whose ulp is 1/4 -- its bits are 0x4A000000 | (H << 2): THE LOW 24 BITS ARE THE INTEGER
  //            4 H (every partial sum is an integer below 2^24, so the f32 accumulation is exact in any order).
  //   rows     D row m = 4 q + j of M-block mb is patch row 7 q + 4 mb + j: lane group q = lane >> 4 ends up holding the 16
  //            consecutive rows 7 q .. 7 q + 15 of its blur column n = lane & 15 -- all the vertical pass needs for the 7
  //            blur rows 7 q .. 7 q + 6.  (Rows 33 .. 36 are uninitialised LDS: finite after the OR, and only feed outputs
  //            nobody reads.)
  //   v-pass   7 x 7 v_mad_u32_u24 straight on the accumulator bits (the 24-bit multiplier reads exactly 4 H): the rounded
  //            result is the top byte of sum 4 H g + 2^23, as in the VALU variant.
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int n16 = lane & 15, q4 = lane >> 4;
  const f16x8 bfrag = __builtin_bit_cast(f16x8, bw);
  // A: patch row 7 ((lane & 15) >> 2) + (lane & 3) + 4 mb, bytes 16 nb + 8 (lane >> 4) .. + 7
  const uint32_t* arow = reinterpret_cast<const uint32_t*>(patch) + (7 * (n16 >> 2) + (lane & 3)) * kRowDw + 2 * q4;
  const uint32_t k64 = 0x64646464u;
  uint32_t hs[2][16];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const uint32_t d0 = arow[4 * mb * kRowDw + 4 * nb], d1 = arow[4 * mb * kRowDw + 4 * nb + 1];
      uint4 af;
      af.x = __builtin_amdgcn_perm(k64, d0, 0x04010400u);  // {1024 + p0, 1024 + p1}
      af.y = __builtin_amdgcn_perm(k64, d0, 0x04030402u);
      af.z = __builtin_amdgcn_perm(k64, d1, 0x04010400u);
      af.w = __builtin_amdgcn_perm(k64, d1, 0x04030402u);
      const f32x4 c0 = {0.0f, 0.0f, 0.0f, 0.0f};  // (an inline constant: no accumulator set-up)
      const f32x4 dd = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af), bfrag, c0, 0, 0, 0);
      // (__float_as_uint of an rvalue: __builtin_bit_cast applied to a vector ELEMENT reads element 0 whatever the index)
      hs[nb][4 * mb + 0] = __float_as_uint(dd[0]);
      hs[nb][4 * mb + 1] = __float_as_uint(dd[1]);
      hs[nb][4 * mb + 2] = __float_as_uint(dd[2]);
      hs[nb][4 * mb + 3] = __float_as_uint(dd[3]);
    }
  // The 49 multiply-adds of a column block are ONE asm statement that opens with its own wait states: the compiler's hazard
  // recogniser does not look inside asm statements, and a VALU read of an MFMA result needs up to 18 of them (an asm
  // v_mad_u32_u24 per term read the accumulators while the matrix core was still writing them; written with __umul24 the
  // compiler emits v_mul_u32_u24 pairs + v_add3_u32 instead -- 136 instructions for these 98).  (The factor 4 of the VALU
  // variant's weights is in the accumulator bits.)
  const uint32_t gw0 = 144u, gw1 = 268u, gw2 = 391u, gw3 = 442u, c23 = 1u << 23;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    uint32_t o[8];
    asm volatile("s_nop 15\n\ts_nop 2\n\tv_mad_u32_u24 %0, %7, %20, %24\n\tv_mad_u32_u24 %0, %8, %21, %0\n\tv_mad_u32_u24 %0, %9, %22, %0\n\tv_mad_u32_u24 %0, %10, %23, %0\n\tv_mad_u32_u24 %0, %11, %22, %0\n\tv_mad_u32_u24 %0, %12, %21, %0\n\tv_mad_u32_u24 %0, %13, %20, %0\n\tv_mad_u32_u24 %1, %8, %20, %24\n\tv_mad_u32_u24 %1, %9, %21, %1\n\tv_mad_u32_u24 %1, %10, %22, %1\n\tv_mad_u32_u24 %1, %11, %23, %1\n\tv_mad_u32_u24 %1, %12, %22, %1\n\tv_mad_u32_u24 %1, %13, %21, %1\n\tv_mad_u32_u24 %1, %14, %20, %1\n\tv_mad_u32_u24 %2, %9, %20, %24\n\tv_mad_u32_u24 %2, %10, %21, %2\n\tv_mad_u32_u24 %2, %11, %22, %2\n\tv_mad_u32_u24 %2, %12, %23, %2\n\tv_mad_u32_u24 %2, %13, %22, %2\n\tv_mad_u32_u24 %2, %14, %21, %2\n\tv_mad_u32_u24 %2, %15, %20, %2\n\tv_mad_u32_u24 %3, %10, %20, %24\n\tv_mad_u32_u24 %3, %11, %21, %3\n\tv_mad_u32_u24 %3, %12, %22, %3\n\tv_mad_u32_u24 %3, %13, %23, %3\n\tv_mad_u32_u24 %3, %14, %22, %3\n\tv_mad_u32_u24 %3, %15, %21, %3\n\tv_mad_u32_u24 %3, %16, %20, %3\n\tv_mad_u32_u24 %4, %11, %20, %24\n\tv_mad_u32_u24 %4, %12, %21, %4\n\tv_mad_u32_u24 %4, %13, %22, %4\n\tv_mad_u32_u24 %4, %14, %23, %4\n\tv_mad_u32_u24 %4, %15, %22, %4\n\tv_mad_u32_u24 %4, %16, %21, %4\n\tv_mad_u32_u24 %4, %17, %20, %4\n\tv_mad_u32_u24 %5, %12, %20, %24\n\tv_mad_u32_u24 %5, %13, %21, %5\n\tv_mad_u32_u24 %5, %14, %22, %5\n\tv_mad_u32_u24 %5, %15, %23, %5\n\tv_mad_u32_u24 %5, %16, %22, %5\n\tv_mad_u32_u24 %5, %17, %21, %5\n\tv_mad_u32_u24 %5, %18, %20, %5\n\tv_mad_u32_u24 %6, %13, %20, %24\n\tv_mad_u32_u24 %6, %14, %21, %6\n\tv_mad_u32_u24 %6, %15, %22, %6\n\tv_mad_u32_u24 %6, %16, %23, %6\n\tv_mad_u32_u24 %6, %17, %22, %6\n\tv_mad_u32_u24 %6, %18, %21, %6\n\tv_mad_u32_u24 %6, %19, %20, %6"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6])
                 : "v"(hs[nb][0]), "v"(hs[nb][1]), "v"(hs[nb][2]), "v"(hs[nb][3]), "v"(hs[nb][4]), "v"(hs[nb][5]), "v"(hs[nb][6]),
                   "v"(hs[nb][7]), "v"(hs[nb][8]), "v"(hs[nb][9]), "v"(hs[nb][10]), "v"(hs[nb][11]), "v"(hs[nb][12]),
                   "v"(gw0), "v"(gw1), "v"(gw2), "v"(gw3), "s"(c23));
    o[7] = 0u;
    uint2 pk;
    pk.x = __builtin_amdgcn_perm(__builtin_amdgcn_perm(o[3], o[2], 0x0c0c0703u), __builtin_amdgcn_perm(o[1], o[0], 0x0c0c0703u), 0x05040100u);
    pk.y = __builtin_amdgcn_perm(__builtin_amdgcn_perm(o[7], o[6], 0x0c0c0703u), __builtin_amdgcn_perm(o[5], o[4], 0x0c0c0703u), 0x05040100u);
    // blur column 16 nb + n16 (columns 27 .. 31 are scratch lines of the 1 KB buffer), rows 7 q4 .. 7 q4 + 6
    *reinterpret_cast<uint2*>(bl + (16 * nb + n16) * 32 + 8 * q4) = pk;
  }
}

// The same for the 45 x 45 patch of the continuous-steering mode (describe_kernel<19, true, true>): rows of 12 dwords, patch column
// 0 at byte 0, 49 rows allocated.  Three 16-column blocks (blur columns 0 .. 38 + scratch), each with its own K window (patch
// columns 16 nb .. 16 nb + 31: past column 47 the window runs into the next row -- finite after the OR, under zero taps).  D row
// m = 4 q + j of M-block mb is patch row 10 q + 4 mb + j: lane group q holds the 16 consecutive rows 10 q .. 10 q + 15 of its blur
// column, enough for the TEN blur rows 10 q .. 10 q + 9 (4 x 10 = 40 >= 39).  Blurred patch: column-major, 48 bytes per column,
// the ten rows of group q at byte 12 q (blur_offset_mfma19).
__host__ __device__ constexpr int blur_offset_mfma19(int row, int col) { return col * 48 + 12 * (row / 10) + row % 10; }
__device__ __forceinline__ void desc_blur_mfma19(const uint8_t* patch, uint8_t* bl, int lane, const uint4 bw) {
  constexpr int kRowDw = 12;
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int n16 = lane & 15, q4 = lane >> 4;
  const f16x8 bfrag = __builtin_bit_cast(f16x8, bw);
  const uint32_t* arow = reinterpret_cast<const uint32_t*>(patch) + (10 * (n16 >> 2) + (lane & 3)) * kRowDw + 2 * q4;
  const uint32_t k64 = 0x64646464u;
  const uint32_t gw0 = 144u, gw1 = 268u, gw2 = 391u, gw3 = 442u, c23 = 1u << 23;
#pragma unroll
  for (int nb = 0; nb < 3; ++nb) {
    uint32_t hs[16];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const uint32_t d0 = arow[4 * mb * kRowDw + 4 * nb], d1 = arow[4 * mb * kRowDw + 4 * nb + 1];
      uint4 af;
      af.x = __builtin_amdgcn_perm(k64, d0, 0x04010400u);  // {1024 + p0, 1024 + p1}
      af.y = __builtin_amdgcn_perm(k64, d0, 0x04030402u);
      af.z = __builtin_amdgcn_perm(k64, d1, 0x04010400u);
      af.w = __builtin_amdgcn_perm(k64, d1, 0x04030402u);
      const f32x4 c0 = {0.0f, 0.0f, 0.0f, 0.0f};
      const f32x4 dd = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af), bfrag, c0, 0, 0, 0);
      hs[4 * mb + 0] = __float_as_uint(dd[0]);
      hs[4 * mb + 1] = __float_as_uint(dd[1]);
      hs[4 * mb + 2] = __float_as_uint(dd[2]);
      hs[4 * mb + 3] = __float_as_uint(dd[3]);
    }
    // two asm statements of 35 multiply-adds (an asm statement takes at most 30 operands): blur rows 0 .. 4 from h rows 0 .. 10,
    // rows 5 .. 9 from h rows 5 .. 15; the first opens with the wait states a VALU read of an MFMA result needs (see desc_blur_mfma)
    uint32_t o[12];
    asm volatile("s_nop 15\n\ts_nop 2\n\tv_mad_u32_u24 %0, %5, %16, %20\n\tv_mad_u32_u24 %0, %6, %17, %0\n\tv_mad_u32_u24 %0, %7, %18, %0\n\tv_mad_u32_u24 %0, %8, %19, %0\n\tv_mad_u32_u24 %0, %9, %18, %0\n\tv_mad_u32_u24 %0, %10, %17, %0\n\tv_mad_u32_u24 %0, %11, %16, %0\n\tv_mad_u32_u24 %1, %6, %16, %20\n\tv_mad_u32_u24 %1, %7, %17, %1\n\tv_mad_u32_u24 %1, %8, %18, %1\n\tv_mad_u32_u24 %1, %9, %19, %1\n\tv_mad_u32_u24 %1, %10, %18, %1\n\tv_mad_u32_u24 %1, %11, %17, %1\n\tv_mad_u32_u24 %1, %12, %16, %1\n\tv_mad_u32_u24 %2, %7, %16, %20\n\tv_mad_u32_u24 %2, %8, %17, %2\n\tv_mad_u32_u24 %2, %9, %18, %2\n\tv_mad_u32_u24 %2, %10, %19, %2\n\tv_mad_u32_u24 %2, %11, %18, %2\n\tv_mad_u32_u24 %2, %12, %17, %2\n\tv_mad_u32_u24 %2, %13, %16, %2\n\tv_mad_u32_u24 %3, %8, %16, %20\n\tv_mad_u32_u24 %3, %9, %17, %3\n\tv_mad_u32_u24 %3, %10, %18, %3\n\tv_mad_u32_u24 %3, %11, %19, %3\n\tv_mad_u32_u24 %3, %12, %18, %3\n\tv_mad_u32_u24 %3, %13, %17, %3\n\tv_mad_u32_u24 %3, %14, %16, %3\n\tv_mad_u32_u24 %4, %9, %16, %20\n\tv_mad_u32_u24 %4, %10, %17, %4\n\tv_mad_u32_u24 %4, %11, %18, %4\n\tv_mad_u32_u24 %4, %12, %19, %4\n\tv_mad_u32_u24 %4, %13, %18, %4\n\tv_mad_u32_u24 %4, %14, %17, %4\n\tv_mad_u32_u24 %4, %15, %16, %4"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4])
                 : "v"(hs[0]), "v"(hs[1]), "v"(hs[2]), "v"(hs[3]), "v"(hs[4]), "v"(hs[5]), "v"(hs[6]), "v"(hs[7]), "v"(hs[8]), "v"(hs[9]),
                   "v"(hs[10]), "v"(gw0), "v"(gw1), "v"(gw2), "v"(gw3), "s"(c23));
    asm volatile("v_mad_u32_u24 %0, %5, %16, %20\n\tv_mad_u32_u24 %0, %6, %17, %0\n\tv_mad_u32_u24 %0, %7, %18, %0\n\tv_mad_u32_u24 %0, %8, %19, %0\n\tv_mad_u32_u24 %0, %9, %18, %0\n\tv_mad_u32_u24 %0, %10, %17, %0\n\tv_mad_u32_u24 %0, %11, %16, %0\n\tv_mad_u32_u24 %1, %6, %16, %20\n\tv_mad_u32_u24 %1, %7, %17, %1\n\tv_mad_u32_u24 %1, %8, %18, %1\n\tv_mad_u32_u24 %1, %9, %19, %1\n\tv_mad_u32_u24 %1, %10, %18, %1\n\tv_mad_u32_u24 %1, %11, %17, %1\n\tv_mad_u32_u24 %1, %12, %16, %1\n\tv_mad_u32_u24 %2, %7, %16, %20\n\tv_mad_u32_u24 %2, %8, %17, %2\n\tv_mad_u32_u24 %2, %9, %18, %2\n\tv_mad_u32_u24 %2, %10, %19, %2\n\tv_mad_u32_u24 %2, %11, %18, %2\n\tv_mad_u32_u24 %2, %12, %17, %2\n\tv_mad_u32_u24 %2, %13, %16, %2\n\tv_mad_u32_u24 %3, %8, %16, %20\n\tv_mad_u32_u24 %3, %9, %17, %3\n\tv_mad_u32_u24 %3, %10, %18, %3\n\tv_mad_u32_u24 %3, %11, %19, %3\n\tv_mad_u32_u24 %3, %12, %18, %3\n\tv_mad_u32_u24 %3, %13, %17, %3\n\tv_mad_u32_u24 %3, %14, %16, %3\n\tv_mad_u32_u24 %4, %9, %16, %20\n\tv_mad_u32_u24 %4, %10, %17, %4\n\tv_mad_u32_u24 %4, %11, %18, %4\n\tv_mad_u32_u24 %4, %12, %19, %4\n\tv_mad_u32_u24 %4, %13, %18, %4\n\tv_mad_u32_u24 %4, %14, %17, %4\n\tv_mad_u32_u24 %4, %15, %16, %4"
                 : "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]), "=&v"(o[9])
                 : "v"(hs[5]), "v"(hs[6]), "v"(hs[7]), "v"(hs[8]), "v"(hs[9]), "v"(hs[10]), "v"(hs[11]), "v"(hs[12]), "v"(hs[13]),
                   "v"(hs[14]), "v"(hs[15]), "v"(gw0), "v"(gw1), "v"(gw2), "v"(gw3), "s"(c23));
    // the top bytes of the ten sums -> 10 bytes of the column: three dword stores (the group's 12-byte slot)
    const uint32_t w0 = __builtin_amdgcn_perm(__builtin_amdgcn_perm(o[3], o[2], 0x0c0c0703u), __builtin_amdgcn_perm(o[1], o[0], 0x0c0c0703u), 0x05040100u);
    const uint32_t w1 = __builtin_amdgcn_perm(__builtin_amdgcn_perm(o[7], o[6], 0x0c0c0703u), __builtin_amdgcn_perm(o[5], o[4], 0x0c0c0703u), 0x05040100u);
    const uint32_t w2 = __builtin_amdgcn_perm(o[9], o[8], 0x0c0c0703u);
    uint32_t* dst = reinterpret_cast<uint32_t*>(bl + (16 * nb + n16) * 48 + 12 * q4);
    dst[0] = w0;
    dst[1] = w1;
    dst[2] = w2;
  }
}

template <int BR, bool STEER, bool MF = false>
__global__ __launch_bounds__(256, (MF && BR == 19) ? 8 : 1) void describe_kernel(DescribeArgs a, DevTables tb, int K,
                                                       const SelKp* __restrict__ sel,
                                                       const int32_t* __restrict__ level_cnt,
                                                       gh_keypoint* __restrict__ kps, uint8_t* __restrict__ desc,
                                                       int32_t* __restrict__ counts, int n_frames,
                                                       uint32_t* __restrict__ dbg) {
  typedef DescGeom<BR> G;
  constexpr int kC = G::kC, kPatch = G::kPatch, kPatchPitch = G::kPatchPitch, kRowDw = G::kRowDw, kBlur = G::kBlur,
                kBlurPitch = G::kBlurPitch, kGroups = G::kGroups;
  static_assert(kPatch <= 64 && kC >= 16, "one lane per patch row; the radius-15 centroid disc lies inside the patch");
  static_assert(!MF || (BR == 13 && !STEER) || (BR == 19 && STEER), "the MFMA blur is written for the 33 x 33 patch of the table mode and the 45 x 45 one of the continuous mode");
  // MF: four more (uninitialised) rows below the patch -- the row map of the MFMA h-pass runs to row 36 with constant offsets
  constexpr int kPatchRows = MF ? kPatch + 4 : kPatch;
  __shared__ __attribute__((aligned(16))) uint8_t s_patch[4][kPatchRows * kPatchPitch + 28];  // + slack for the 16-B row reads
  __shared__ __attribute__((aligned(16))) uint32_t s_h[4][MF ? (BR == 13 ? 256 : 48 * 48 / 4) : (kPatch + 1) * kBlurPitch];  // MF: the blurred patch (1 KB / 2.25 KB)
  // VALU variant: the blurred patch REPLACES the raw one (last read by the h-pass, a wave barrier before the v-pass writes): 20.2 KB of
  // LDS per workgroup in the table mode = 8 workgroups per CU instead of 6 (GSLAM_HIP_ORB_DESC_LDSPAD=3000 restores 6 for A/B runs)
  static_assert(sizeof(s_patch[0]) >= kBlur * kBlurPitch + 12, "the blurred patch fits where the raw patch was");
  static_assert((BR != 13 && !MF) || sizeof(s_patch) + sizeof(s_h) <= 20480, "8 workgroups per CU");
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int blocks_per_frame = (K + 3) >> 2;
  const int gid = xcd_strip_tile(blockIdx.x, blocks_per_frame * n_frames);
  if (gid >= blocks_per_frame * n_frames) return;
  // frame and slot are the same for the whole wave: through v_readfirstlane so that everything derived from them (level,
  // output position, every base address) is scalar arithmetic instead of quarter-rate 64-bit VALU multiplies
  const int b = __builtin_amdgcn_readfirstlane(gid / blocks_per_frame);
  const int slot = __builtin_amdgcn_readfirstlane((gid - b * blocks_per_frame) * 4 + wv);
  if (slot >= K) return;
  // level of this slot and compacted output position
  int l = 0;
  for (int k = 1; k < a.nlevels; ++k)
    if (slot >= a.quota_off[k]) l = k;
  const int i = slot - a.quota_off[l];
  int before = 0, unused_before = 0, total = 0;
  for (int k = 0; k < a.nlevels; ++k) {
    const int c = level_cnt[b * kMaxL + k];
    total += c;
    if (k < l) {
      before += c;
      unused_before += a.quota[k] - c;
    }
  }
  const int cnt_l = level_cnt[b * kMaxL + l];
  if (slot == 0 && lane == 0) counts[b] = total;
  if (i >= cnt_l) {  // unused slot: zero-fill one tail row so the whole K-row output is deterministic
    const int pos = total + unused_before + (i - cnt_l);
    if (dbg != nullptr && lane == 0) atomicAdd(&dbg[kDbgUnusedSlots], 1u);
    if (pos < K) {
      if (lane < 7) reinterpret_cast<uint32_t*>(kps + (size_t)b * K + pos)[lane] = 0u;
      if (lane >= 8 && lane < 16) reinterpret_cast<uint32_t*>(desc + ((size_t)b * K + pos) * 32)[lane - 8] = 0u;
    }
    return;
  }
  const int pos = before + i;
  const SelKp kp = sel[(size_t)b * K + slot];
  const LevelView lv = a.lv[l];
  const uint8_t* img = lv.base + (size_t)b * lv.frame_stride;
  // kPatch x kPatch patch: each row is fetched as the kRowDw aligned dwords that cover it; the wanted bytes start at
  // offset px0 & 3 (= px0 - pa) of the row in LDS.
  const int px0 = (int)kp.x - kC, py0 = (int)kp.y - kC;
  const int pa = px0 & ~3;
  // lane r < kPatch fetches the whole row r (dwordx4 loads + a tail: 3 vector-memory instructions per wave instead of
  // trips of address arithmetic + dword loads).  The table mode's patch (radius 16) never leaves the image (keypoints keep
  // 19 px from the border); the radius-22 patch of the continuous mode can: such keypoints gather their patch byte by
  // byte with BORDER_REFLECT_101 coordinates (what ORB-SLAM's padded pyramid holds there).
  const bool inside = !STEER || (pa >= 0 && pa + kPatchPitch <= lv.pitch && px0 + kPatch <= lv.w && py0 >= 0 && py0 + kPatch <= lv.h);
  if (lane < kPatch) {
    uint32_t* dst = reinterpret_cast<uint32_t*>(&s_patch[wv][lane * kPatchPitch]);
    if (inside) {
      struct __attribute__((packed, aligned(4))) RowN { uint32_t w[kRowDw]; };
      // (rows and pitch are < 2^24 and a level is < 4 GiB: a 32-bit offset on the full-rate 24-bit multiplier)
      const RowN row = *reinterpret_cast<const RowN*>(img + (__umul24((uint32_t)(py0 + lane), (uint32_t)lv.pitch) + (uint32_t)pa));
      if constexpr (MF) {
        // the MFMA operands are read as whole dwords: shift the row so that patch column 0 is byte 0 of its LDS row
        const uint32_t sh = (uint32_t)(px0 - pa);
#pragma unroll
        for (int c = 0; c < kRowDw; ++c) dst[c] = __builtin_amdgcn_alignbyte(c + 1 < kRowDw ? row.w[c + 1] : 0u, row.w[c], sh);
      } else {
#pragma unroll
        for (int c = 0; c < kRowDw; ++c) dst[c] = row.w[c];
      }
    } else {
      const uint8_t* rp = img + (size_t)orb_reflect101(py0 + lane, lv.h) * lv.pitch;
      for (int c = 0; c < kRowDw; ++c) {
        uint32_t w = 0;
        for (int e = 0; e < 4; ++e) w |= (uint32_t)rp[orb_reflect101((MF ? px0 : pa) + 4 * c + e, lv.w)] << (8 * e);
        dst[c] = w;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  int m10, m01;
  desc_moments<kC, kRowDw>(s_patch[wv], MF ? 0u : (uint32_t)(px0 - pa), lane, &m10, &m01);
  int bin = 0;
  float angle = 0.0f;
  if constexpr (STEER) {
    angle = orb_fast_atan2_deg((float)m01, (float)m10);  // |m| < 2^24: the conversions are exact
  } else {
    bin = desc_bin(tb.dir, m10, m01, lane);
    angle = 12.0f * (float)bin;
  }
  // the four test words of this lane, requested before the blur so that their latency (an L2 hit each) hides under it instead
  // of forming a chain of four load -> test -> store round trips behind it
  const uint32_t* pat = STEER ? reinterpret_cast<const uint32_t*>(tb.base_pattern)
                              : reinterpret_cast<const uint32_t*>(tb.pattern) + (size_t)bin * 256;
  const uint32_t pws[4] = {pat[lane], pat[64 + lane], pat[128 + lane], pat[192 + lane]};
  uint8_t* bl = MF ? reinterpret_cast<uint8_t*>(s_h[wv]) : s_patch[wv];
  if constexpr (MF && BR == 13) {
    desc_blur_mfma(s_patch[wv], bl, lane, desc_blur_b(lane));
  } else if constexpr (MF) {
    desc_blur_mfma19(s_patch[wv], bl, lane, desc_blur_b(lane));
  } else {
    // separable 7x7 integer Gaussian: patch rows 0..kPatch-1 x blur cols -> s_h, then blur rows -> the blurred patch
    uint32_t* hb = s_h[wv];
    constexpr int kRowsPerTrip = 64 / kGroups;
    const int lrow = lane / kGroups, lgrp = lane - lrow * kGroups;
    constexpr uint32_t g[7] = {144, 268, 391, 442, 391, 268, 144};
    // Wide LDS accesses (the kernel is LDS-issue bound with byte reads): one work item = 4 adjacent outputs.
    // h-pass: 4 dwords of the patch row -> 10 source bytes (v_alignbyte with the wave-uniform row offset)
    //         -> 4 outputs stored as one 16-byte write;  v-pass: 7 x 16-byte reads down the 4 columns -> 4 outputs.
    {
      const uint32_t off = (uint32_t)(px0 - pa);  // 0..3, wave-uniform
      const uint32_t* p32 = reinterpret_cast<const uint32_t*>(s_patch[wv]);
      // lane -> (row of the trip, 4-column group), fixed for the whole kernel: kRowsPerTrip rows x kGroups groups per trip (63 / 60
      // of the 64 lanes; same trip counts as a flat index, without a division by 7 / 10 in every trip)
      for (int r = lrow; r < kPatch && lane < kRowsPerTrip * kGroups; r += kRowsPerTrip) {
        const int gq = lgrp;  // outputs: blur cols 4 gq .. 4 gq + 3 of patch row r
        const uint32_t* q = p32 + r * kRowDw + gq;
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3];
        const uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, off);  // source bytes 0..3 (patch col 4 gq + k)
        const uint32_t w1 = __builtin_amdgcn_alignbyte(d2, d1, off);  // 4..7
        const uint32_t w2 = __builtin_amdgcn_alignbyte(d3, d2, off);  // 8..11
        // P[s] = {x[s], x[s+1]} as two 16-bit lanes (one v_perm each, straight from the 12-byte window); an output is
        // 4 v_dot2_u32_u16 with the tap pairs (g0,g1) (g2,g3) (g4,g5) (g6,0) instead of 7 multiply-adds
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
        u16x2 P[10];
  #pragma unroll
        for (int sft = 0; sft < 10; ++sft) {
          // bytes sft, sft + 1 of {w0, w1, w2}: the perm sees 8 of the 12 bytes
          const uint32_t lo_dw = sft < 7 ? w0 : w1, hi_dw = sft < 7 ? w1 : w2;
          const uint32_t bsel = (uint32_t)(sft < 7 ? sft : sft - 4);
          P[sft] = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(hi_dw, lo_dw, 0x0c000c00u | ((bsel + 1u) << 16) | bsel));
        }
        constexpr uint32_t g01 = 144u | (268u << 16), g23 = 391u | (442u << 16), g45 = 391u | (268u << 16), g6 = 144u;
        uint4 o;
        uint32_t acc[4];
  #pragma unroll
        for (int i = 0; i < 4; ++i) {
          uint32_t t0 = __builtin_amdgcn_udot2(P[i], __builtin_bit_cast(u16x2, g01), 0u, false);
          t0 = __builtin_amdgcn_udot2(P[i + 2], __builtin_bit_cast(u16x2, g23), t0, false);
          t0 = __builtin_amdgcn_udot2(P[i + 4], __builtin_bit_cast(u16x2, g45), t0, false);
          acc[i] = __builtin_amdgcn_udot2(P[i + 6], __builtin_bit_cast(u16x2, g6), t0, false);
        }
        o.x = acc[0]; o.y = acc[1]; o.z = acc[2]; o.w = acc[3];
        *reinterpret_cast<uint4*>(&hb[r * kBlurPitch + 4 * gq]) = o;
      }
    }
    __builtin_amdgcn_wave_barrier();
    // v-pass: one work item = 4 adjacent outputs of one blur row: 7 x 16-byte reads down the 4 columns, 28
    // v_mad_u32_u24 (h sums < 2^20), weights x4 so that the rounded result is the top byte of the sum
    // ((4 s + 2^23) >> 24 == (s + 2^21) >> 22; 4 * 2048 * 522240 + 2^23 < 2^32), one dword store.
    for (int rb = lrow; rb < kBlur && lane < kRowsPerTrip * kGroups; rb += kRowsPerTrip) {
      const int cg = lgrp;
      uint32_t acc[4] = {1u << 23, 1u << 23, 1u << 23, 1u << 23};
  #pragma unroll
      for (int t = 0; t < 7; ++t) {
        const uint4 hv = *reinterpret_cast<const uint4*>(&hb[(rb + t) * kBlurPitch + 4 * cg]);
        acc[0] += __umul24(4u * g[t], hv.x);
        acc[1] += __umul24(4u * g[t], hv.y);
        acc[2] += __umul24(4u * g[t], hv.z);
        acc[3] += __umul24(4u * g[t], hv.w);
      }
      *reinterpret_cast<uint32_t*>(&bl[rb * kBlurPitch + 4 * cg]) =
          __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0703u) | (__builtin_amdgcn_perm(acc[3], acc[2], 0x0c0c0703u) << 16);
    }
  }
  __builtin_amdgcn_wave_barrier();
  // 256 binary tests, 64 per ballot
  uint8_t* drow = desc + ((size_t)b * K + pos) * 32;
  float cs = 1.0f, sn = 0.0f;
  if constexpr (STEER) orb_sincos_deg(angle, &cs, &sn);
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const uint32_t pw = pws[gq];
    int va, vb;
    if constexpr (STEER) {
      // pw = the unrotated test (ax, ay, bx, by) as four int8; (x', y') = (rint(x cos - y sin), rint(x sin + y cos)), ties to even
      const float ax = (float)(int)(int8_t)(pw & 0xFFu), ay = (float)(int)(int8_t)((pw >> 8) & 0xFFu);
      const float bx = (float)(int)(int8_t)((pw >> 16) & 0xFFu), by = (float)(int)(int8_t)(pw >> 24);
      const int rax = (int)rintf(ax * cs - ay * sn), ray = (int)rintf(ax * sn + ay * cs);
      const int rbx = (int)rintf(bx * cs - by * sn), rby = (int)rintf(bx * sn + by * cs);
      if constexpr (MF) {
        // blur_offset_mfma19: column-major, row r of a column at byte r + 2 (r / 10)  (r / 10 = 205 r >> 11 for r < 64)
        const uint32_t ra = (uint32_t)(BR + ray), rb = (uint32_t)(BR + rby);
        va = bl[__umul24((uint32_t)(BR + rax), 48u) + ra + 2u * (__umul24(ra, 205u) >> 11)];
        vb = bl[__umul24((uint32_t)(BR + rbx), 48u) + rb + 2u * (__umul24(rb, 205u) >> 11)];
      } else {
        va = bl[(BR + ray) * kBlurPitch + BR + rax];
        vb = bl[(BR + rby) * kBlurPitch + BR + rbx];
      }
    } else {
      va = bl[pw & 0xFFFFu];  // byte offsets of the two sample points in the blurred patch
      vb = bl[pw >> 16];
    }
    const uint64_t bits = __ballot(va < vb);
    if (lane == 0) *reinterpret_cast<uint64_t*>(drow + 8 * gq) = bits;
  }
  if (lane == 0) {
    const float sc = a.scale[l];
    gh_keypoint o;
    o.x = __fmul_rn((float)kp.x, sc);
    o.y = __fmul_rn((float)kp.y, sc);
    o.size = __fmul_rn(31.0f, sc);
    o.angle = angle;
    o.response = (float)kp.score;
    o.octave = l;
    o.class_id = -1;
    kps[(size_t)b * K + pos] = o;
  }
}

// orb_describe of the default mode (30-bin table steering, blur on MFMA) as a SOFTWARE PIPELINE over the keypoints of a wave.
// describe_kernel is one keypoint per wave and is bound by the latency of its dependent chain (selection record -> patch rows
// from HBM -> compute) at the hardware's 8 waves per SIMD; here a wave owns kDescPipe slots (slot0 + 4 j) and, while it
// computes keypoint j, the patch rows of keypoint j + 1 are already in flight into registers and the selection record of
// keypoint j + 2 into scalar registers.  Same arithmetic, same outputs (desc_moments / desc_bin / desc_blur_mfma).
constexpr int kDescPipe = 8;

__global__ __launch_bounds__(256) void describe_pipe_kernel(DescribeArgs a, DevTables tb, int K, const SelKp* __restrict__ sel,
                                                            const int32_t* __restrict__ level_cnt,
                                                            gh_keypoint* __restrict__ kps, uint8_t* __restrict__ desc,
                                                            int32_t* __restrict__ counts, int n_frames,
                                                            uint32_t* __restrict__ dbg) {
  typedef DescGeom<13> G;
  constexpr int kC = G::kC, kPatch = G::kPatch, kPatchPitch = G::kPatchPitch, kRowDw = G::kRowDw;
  static_assert(kRowDw == 9 && kPatch == 33, "desc_blur_mfma's patch geometry");
  __shared__ __attribute__((aligned(16))) uint8_t s_patch[4][(kPatch + 4) * kPatchPitch + 28];
  __shared__ __attribute__((aligned(16))) uint32_t s_blur[4][256];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // Round 6: a workgroup's slots lie in ONE level's quota (the host deals workgroups per level: DescribeArgs::blk_off), so the level,
  // its view, its scale and the slot arithmetic are wave-uniform loop invariants.  The general form -- any slot of the frame, level
  // and output row found by eight-way selects per slot, the level's view fetched from the argument segment per keypoint -- cost 230
  // SALU instructions per keypoint and several scalar-cache round trips that an in-order wave of this latency-bound kernel (5 waves
  // per SIMD) sits out.
  const int blocks_per_frame = a.blk_off[a.nlevels];
  const int gid = xcd_strip_tile(blockIdx.x, blocks_per_frame * n_frames);
  if (gid >= blocks_per_frame * n_frames) return;
  const int b = __builtin_amdgcn_readfirstlane(gid / blocks_per_frame);
  const int g = __builtin_amdgcn_readfirstlane(gid - b * blocks_per_frame);
  int lvl = 0;
#pragma unroll
  for (int k = 1; k < kMaxL; ++k)
    if (k < a.nlevels && g >= a.blk_off[k]) lvl = k;
  // per-frame level counts, once per wave (scalar loads); what lies in front of this level's rows
  int total = 0, before = 0, unused_before = 0, cnt_l = 0;
#pragma unroll
  for (int k = 0; k < kMaxL; ++k) {
    const int ck = k < a.nlevels ? level_cnt[b * kMaxL + k] : 0;
    total += ck;
    if (k < lvl) {
      before += ck;
      unused_before += a.quota[k] - ck;
    }
    if (k == lvl) cnt_l = ck;
  }
  const int qoff = a.quota_off[lvl], slot_end = qoff + a.quota[lvl];
  const int slot0 = __builtin_amdgcn_readfirstlane(qoff + (g - a.blk_off[lvl]) * (4 * kDescPipe) + wv);
  if (g == 0 && wv == 0 && lane == 0) counts[b] = total;
  if (slot0 >= slot_end) return;
  const LevelView lv_l = a.lv[lvl], lv_0 = a.lv[0];
  const float scale_l = a.scale[lvl];

  // what a slot is: 0 = beyond the level's quota, 1 = unused slot (zero-fill output row pos), 2 = keypoint (output row pos)
  struct SlotInfo { int kind, pos; };
  auto slot_info = [&](int slot) {
    SlotInfo si{0, 0};
    if (slot >= slot_end) return si;
    const int i = slot - qoff;
    si.kind = i >= cnt_l ? 1 : 2;
    si.pos = i >= cnt_l ? total + unused_before + (i - cnt_l) : before + i;
    return si;
  };
  // selection record of a slot as one 8-byte scalar load: x | y << 16, score | level << 8
  // (through the CONSTANT address space: the records were written by an earlier launch, and only such a load is emitted as
  //  s_load inside the loop -- a global load of a uniform address behind this kernel's own stores stays a vector load, whose
  //  in-order vmcnt would make its consumer wait for the prefetched patch rows as well)
  typedef const __attribute__((address_space(4))) uint64_t* sel_cptr;
  const sel_cptr sel_c = (sel_cptr)(reinterpret_cast<const uint64_t*>(sel) + (size_t)b * K);
  auto load_sel = [&](int slot) {
    const uint64_t v = sel_c[slot];
    return uint2{(uint32_t)v, (uint32_t)(v >> 32)};
  };
  // the 9 dwords that cover patch row `lane` of keypoint (x, y) of level l (lanes >= 33: nothing)
  struct __attribute__((packed, aligned(4))) RowN { uint32_t w[kRowDw]; };
  // UNCONDITIONAL, every lane (lanes >= 33 fetch row 32 again; a slot without a keypoint fetches the top-left patch of level
  // 0, which every frame has): a branch around the loads would put them in a block of their own, the compiler's s_waitcnt
  // insertion would merge the paths with and without them conservatively, and the waits for the test words behind them
  // (vmcnt counts in order) would then wait for the prefetched rows too
  auto issue_patch = [&](bool valid, const SlotInfo& si, uint2 raw, RowN& row) {
    const LevelView lv = valid ? lv_l : lv_0;
    const uint32_t xy = valid ? raw.x : (uint32_t)(kC | (kC << 16));
    const uint8_t* img = lv.base + (size_t)b * lv.frame_stride;
    const int px0 = (int)(xy & 0xFFFFu) - kC, py0 = (int)(xy >> 16) - kC;
    const int pa = px0 & ~3;
    row = *reinterpret_cast<const RowN*>(img + (__umul24((uint32_t)(py0 + min(lane, kPatch - 1)), (uint32_t)lv.pitch) + (uint32_t)pa));
  };

  SlotInfo si0 = slot_info(slot0), si1 = slot_info(slot0 + 4);
  uint2 raw0 = si0.kind == 2 ? load_sel(slot0) : uint2{0u, 0u};
  uint2 raw1 = si1.kind == 2 ? load_sel(slot0 + 4) : uint2{0u, 0u};
  RowN row{};
  issue_patch(si0.kind == 2, si0, raw0, row);
  uint8_t* patch = s_patch[wv];
  uint8_t* bl = reinterpret_cast<uint8_t*>(s_blur[wv]);
  // loop invariants held in registers: no vector load inside the loop may sit between the prefetch and its consumer
  const uint4 bw = desc_blur_b(lane);
  const int dir_x = lane < GH_ORB_NBINS ? tb.dir[2 * lane] : 0, dir_y = lane < GH_ORB_NBINS ? tb.dir[2 * lane + 1] : 0;
  for (int j = 0; j < kDescPipe && si0.kind != 0; ++j) {
    // patch rows of keypoint j: registers -> LDS, shifted so that patch column 0 is byte 0 of its row
    if (si0.kind == 2 && lane < kPatch) {
      const uint32_t sh = ((raw0.x & 0xFFFFu) - (uint32_t)kC) & 3u;
      uint32_t* dst = reinterpret_cast<uint32_t*>(patch + lane * kPatchPitch);
#pragma unroll
      for (int c = 0; c < kRowDw; ++c) dst[c] = __builtin_amdgcn_alignbyte(c + 1 < kRowDw ? row.w[c + 1] : 0u, row.w[c], sh);
    }
    __builtin_amdgcn_wave_barrier();
    // keypoint j + 2: its selection record (a scalar load)
    const int slot2 = slot0 + 4 * (j + 2);
    const SlotInfo si2 = j + 2 < kDescPipe ? slot_info(slot2) : SlotInfo{0, 0};
    const uint2 raw2 = si2.kind == 2 ? load_sel(slot2) : uint2{0u, 0u};

    if (si0.kind == 1) {  // unused slot: zero-fill one tail row so the whole K-row output is deterministic
      issue_patch(j + 1 < kDescPipe && si1.kind == 2, si1, raw1, row);
      if (dbg != nullptr && lane == 0) atomicAdd(&dbg[kDbgUnusedSlots], 1u);
      if (si0.pos < K) {
        if (lane < 7) reinterpret_cast<uint32_t*>(kps + (size_t)b * K + si0.pos)[lane] = 0u;
        if (lane >= 8 && lane < 16) reinterpret_cast<uint32_t*>(desc + ((size_t)b * K + si0.pos) * 32)[lane - 8] = 0u;
      }
    } else {
      int m10, m01;
      desc_moments<kC, kRowDw>(patch, 0u, lane, &m10, &m01);
      const int bin = desc_bin(dir_x, dir_y, m10, m01, lane);
      const uint32_t* pat = reinterpret_cast<const uint32_t*>(tb.pattern) + (size_t)bin * 256;
      const uint32_t pws[4] = {pat[lane], pat[64 + lane], pat[128 + lane], pat[192 + lane]};
      // keypoint j + 1: its patch rows go in flight now, BEHIND the test words (vmcnt counts in order: the tests below wait for
      // the words with three loads still outstanding) and ahead of the blur, the tests and the stores
      issue_patch(j + 1 < kDescPipe && si1.kind == 2, si1, raw1, row);
      desc_blur_mfma(patch, bl, lane, bw);
      __builtin_amdgcn_wave_barrier();
      uint8_t* drow = desc + ((size_t)b * K + si0.pos) * 32;
      // all four ballots first, the stores behind them: a (conditional) store between two waits for test words would make the
      // later wait count it as possibly absent and reach into the prefetched rows
      uint64_t bits[4];
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int va = bl[pws[gq] & 0xFFFFu], vb = bl[pws[gq] >> 16];
        bits[gq] = __ballot(va < vb);
      }
      if (lane == 0) {
        reinterpret_cast<uint4*>(drow)[0] = uint4{(uint32_t)bits[0], (uint32_t)(bits[0] >> 32), (uint32_t)bits[1], (uint32_t)(bits[1] >> 32)};
        reinterpret_cast<uint4*>(drow)[1] = uint4{(uint32_t)bits[2], (uint32_t)(bits[2] >> 32), (uint32_t)bits[3], (uint32_t)(bits[3] >> 32)};
        const float sc = scale_l;
        gh_keypoint o;
        o.x = __fmul_rn((float)(raw0.x & 0xFFFFu), sc);
        o.y = __fmul_rn((float)(raw0.x >> 16), sc);
        o.size = __fmul_rn(31.0f, sc);
        o.angle = 12.0f * (float)bin;
        o.response = (float)(raw0.y & 0xFFu);
        o.octave = lvl;
        o.class_id = -1;
        kps[(size_t)b * K + si0.pos] = o;
      }
    }
    si0 = si1; raw0 = raw1;
    si1 = si2; raw1 = raw2;
  }
}

__global__ void bgr_to_gray_kernel(const uint8_t* __restrict__ bgr, int w, int h, int channels, int sstride,
                                   size_t src_frame_stride, uint8_t* __restrict__ gray, int dstride, size_t dst_frame_stride) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w || y >= h) return;
  const uint8_t* p = bgr + (size_t)blockIdx.z * src_frame_stride + (size_t)y * sstride + (size_t)x * channels;
  gray[(size_t)blockIdx.z * dst_frame_stride + (size_t)y * dstride + x] =
      (uint8_t)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14);
}

long long ipow(int b, int e) {
  long long r = 1;
  while (e-- > 0) r *= b;
  return r;
}

}  // namespace

// (host only) the multiplier the FAST launch uses instead of an integer division: see magic_div
extern "C" uint32_t gh_magic_div(uint32_t d, uint32_t n_max) { return magic_div(d, n_max); }

// ------------------------------------------------------------------------------------------------
struct gh_orb_plan {
  gh_ctx* ctx = nullptr;
  int w = 0, h = 0, max_batch = 0, L = 0;
  gh_orb_params prm{};
  int lw[kMaxL]{}, lh[kMaxL]{}, pitch[kMaxL]{}, quota[kMaxL]{}, quota_off[kMaxL]{};
  int ncx[kMaxL]{}, ncy[kMaxL]{}, cell_off[kMaxL]{};
  float scale[kMaxL]{};
  size_t lvl_off[kMaxL]{};  // offset of level l inside one frame's pyramid slab
  size_t slab = 0;
  int cells_per_frame = 0;
  uint8_t* pyr = nullptr;
  uint32_t* xtab[kMaxL]{};
  uint32_t* ytab[kMaxL]{};
  uint32_t* xsel[kMaxL]{};  // per output column: perm selector / weight pair of the horizontal lerp (ResizeTabs)
  uint32_t* xwgt[kMaxL]{};
  uint32_t* mtab[kMaxL]{};   // MFMA resize: A operands per 8-column group of level l / window starts (ResizeTabs); null = not available
  uint32_t* mcw[kMaxL]{};
  bool resize_mfma = true;   // GSLAM_HIP_ORB_RESIZE_MFMA=0: the fused resize of interior tiles on the VALU as well (rounds 2-4)
  uint32_t* cell_cnt = nullptr;
  uint32_t* cell_ent = nullptr;
  SelKp* sel = nullptr;
  int32_t* level_cnt = nullptr;
  const int32_t* own_gx[kMaxL] = {nullptr};  // fused pyramid: first owned output group / row per tile column / row of level l
  const int32_t* own_gy[kMaxL] = {nullptr};
  bool fuse_pyramid = true;  // GSLAM_HIP_ORB_FUSE_PYRAMID=0 keeps the stand-alone resize launches (A/B measurements)
  bool pyramid_ahead = false;  // GSLAM_HIP_ORB_FUSE_PYRAMID=2: the stand-alone resize chain runs AHEAD on its own stream, beside the FAST passes
  hipStream_t pyr_stream = nullptr;
  hipEvent_t ev_pyr[kMaxL]{}, ev_pyr_start = nullptr;
  int desc_lds_pad = 0;      // GSLAM_HIP_ORB_DESC_LDSPAD: the same for orb_describe
  int desc_mfma = 2;         // GSLAM_HIP_ORB_DESC_MFMA: 0 = the 7x7 blur of orb_describe on the VALU (rounds 1-4), 1 = h-pass on MFMA, one keypoint per wave, 2 = MFMA + software pipeline over 8 keypoints per wave
  int lds_pad = 0;           // GSLAM_HIP_ORB_LDSPAD: extra dynamic LDS bytes per workgroup (occupancy experiments only)
  int pass1 = 1;             // GSLAM_HIP_ORB_PASS1: 0 = packed 16-bit compass test (rounds 2-3), 1 = SWAR on 16-bit fields
  bool pk_score = true;      // GSLAM_HIP_ORB_PKSCORE=0: arc scores with v_min3 / v_max3_u32 instead of packed fp16 minimum3 / maximum3
  int8_t* d_pattern = nullptr;
  int8_t* d_base_pattern = nullptr;  // the unrotated tests, 256 x 4 int8 (continuous steering)
  int8_t base_pattern[256 * 4] = {};
  bool base_pattern_fits_table = true;  // every 12-degree rotation stays within +-13 (the 30-bin table exists)
  int distribution = 0;              // gh_orb_plan_set_distribution: 0 = 32 x 32 cells + rank order, 1 = ORB-SLAM's cells + quadtree
  gh_qt_plan* qt = nullptr;          // buffers of mode 1 (orb_quadtree.hip)
  uint8_t* score_plane = nullptr;    // mode 1: S of every level (fast_cells_kernel<.., PLANE>), allocated with qt; GSLAM_HIP_QT_PLANE=0: cells from the image
  size_t plane_off[kMaxL] = {}, plane_slab = 0;
  int plane_pitch[kMaxL] = {};
  int steer = 0;                     // gh_orb_plan_set_steering: 0 = 30 orientation bins, 1 = continuous (fastAtan2 + per-keypoint rotation)
  int32_t* d_dir = nullptr;
  uint32_t* tabs = nullptr;
  uint32_t* dbg = nullptr;  // kDbgCount counters, allocated by gh_orb_plan_debug_counters(enable)
  bool dbg_on = false;
  // host staging for gh_orb_extract_host
  // single-frame host entry point: device staging (image; count | keypoints | descriptors in ONE block so that the
  // results come back in one copy) and a pinned host mirror of the result block
  // small calls are launch-bound (11 launches for a 640x480 frame whose kernels take a few microseconds each): the launch
  // sequence of a call is captured once per argument set and replayed as ONE hipGraph launch (gh_orb_extract_dev)
  struct CallGraph {
    const void *gray, *kps, *desc, *counts;
    int batch, row_stride;
    size_t frame_stride;
    hipGraphExec_t exec;
    hipEvent_t done;  // recorded behind every launch of `exec`: an evicted graph is destroyed only once this has passed
  };
  // A ring of staging slots (gh_orb_stream: depth <= 16) or a caller that reuses its buffers has a handful of argument
  // sets.  A caller that hands in fresh pointers every call (extract() without out=) never hits: after kGraphGiveUp
  // misses that outnumber the hits four to one the plan stops capturing and launches kernel by kernel (a miss costs a
  // capture + an instantiation on top of the launches it saves).
  static constexpr int kGraphRing = 16, kGraphGiveUp = 8;
  std::vector<CallGraph> graphs, retired;
  long long graph_hits = 0, graph_misses = 0;
  hipStream_t cap_stream = nullptr;  // the capture runs on a stream of the plan, never on the caller's (another host thread may be enqueuing there)
  bool graphs_off = false, capturing = false;
  // batched calls: select(level l) runs on a side stream beside fast_cells(l + 1 ..) (gh_orb_extract_dev)
  hipStream_t side = nullptr;
  hipEvent_t ev_level[kMaxL]{}, ev_join = nullptr;
  uint8_t* stage_img = nullptr;
  uint8_t* stage_out = nullptr;
  uint8_t* stage_host = nullptr;  // hipHostMalloc
  size_t stage_img_bytes = 0;
  size_t bytes = 0;
};

extern "C" void gh_orb_default_params(gh_orb_params* p) {
  if (!p) return;
  p->n_features = 1000;
  p->n_levels = 8;
  p->ini_th_fast = 20;
  p->min_th_fast = 7;
}

static gh_status plan_alloc(gh_orb_plan* p, size_t bytes, void** out) {
  gh_status s = gh_dev_alloc(p->ctx, bytes, out);
  if (s == GH_OK) p->bytes += bytes;
  return s;
}

extern "C" void gh_orb_plan_destroy(gh_orb_plan* p) {
  if (!p) return;
  gh_ctx* c = p->ctx;
  GH_ENTER(c);
  hipStreamSynchronize(c->stream);
  void* ptrs[] = {p->pyr, p->cell_cnt, p->cell_ent, p->sel, p->level_cnt, p->d_pattern, p->d_base_pattern, p->d_dir, p->tabs,
                  p->stage_img, p->stage_out, p->dbg};
  for (void* q : ptrs)
    if (q) hipFree(q);
  if (p->stage_host) hipHostFree(p->stage_host);
  gh_qt_destroy(p->qt);
  for (auto* v : {&p->graphs, &p->retired})
    for (auto& g : *v) {
      hipEventSynchronize(g.done);  // (the caller may have moved the context to another stream since the last launch)
      hipGraphExecDestroy(g.exec);
      hipEventDestroy(g.done);
    }
  if (p->cap_stream) hipStreamDestroy(p->cap_stream);
  if (p->pyr_stream) {
    hipStreamSynchronize(p->pyr_stream);
    hipStreamDestroy(p->pyr_stream);
  }
  for (hipEvent_t e : p->ev_pyr)
    if (e) hipEventDestroy(e);
  if (p->ev_pyr_start) hipEventDestroy(p->ev_pyr_start);
  if (p->side) {
    hipStreamSynchronize(p->side);
    hipStreamDestroy(p->side);
  }
  for (hipEvent_t e : p->ev_level)
    if (e) hipEventDestroy(e);
  if (p->ev_join) hipEventDestroy(p->ev_join);
  delete p;
}

// The device copy of the test pattern holds, per (bin, test), the two byte offsets into the 27 x 28 blurred patch
// ((13 + y) * kBlurPitch + 13 + x as two u16) instead of the four int8 coordinates: saves the sign extensions and
// address arithmetic of 256 tests per keypoint.  rot = [30][256][4] rotated coordinates, each within +-13.
static gh_status upload_pattern(gh_orb_plan* p, const int8_t* rot) {
  static_assert(sizeof(GH_ORB_PATTERN) == 30 * 256 * 4, "pattern table layout");
  std::vector<uint32_t> off(30 * 256);
  for (int b = 0; b < 30; ++b)
    for (int t = 0; t < 256; ++t) {
      const int8_t* q = rot + ((size_t)b * 256 + t) * 4;
      const uint32_t oa = p->desc_mfma ? (uint32_t)blur_offset_mfma(13 + q[1], 13 + q[0]) : (uint32_t)((13 + q[1]) * kBlurPitch + 13 + q[0]);
      const uint32_t ob = p->desc_mfma ? (uint32_t)blur_offset_mfma(13 + q[3], 13 + q[2]) : (uint32_t)((13 + q[3]) * kBlurPitch + 13 + q[2]);
      off[b * 256 + t] = oa | (ob << 16);
    }
  return gh_dev_upload(p->ctx, p->d_pattern, off.data(), off.size() * sizeof(uint32_t));
}

// Steered-BRIEF look-up table of a caller-supplied test pattern (Rublee et al. 2011, section 4.2): bin k rotates every
// point by 12 k degrees, (x', y') = (round(x cos - y sin), round(x sin + y cos)), round half away from zero -- the rule
// tools/gen_orb_tables.py used for the built-in pattern, so handing the built-in bin-0 pattern back reproduces it.
extern "C" gh_status gh_orb_plan_set_pattern(gh_orb_plan* p, const int8_t* pattern) {
  if (!p) return GH_ERR_ARG;
  gh_ctx* ctx = p->ctx;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, pattern != nullptr);
  std::vector<int8_t> rot((size_t)30 * 256 * 4);
  auto rnd = [](double v) { return (int)floor(fabs(v) + 0.5) * (v >= 0 ? 1 : -1); };
  bool fits_table = true;
  int bad_t = -1, bad_k = 0;
  for (int t = 0; t < 256; ++t) {
    const int8_t* q = pattern + 4 * t;
    if (q[0] == q[2] && q[1] == q[3]) return gh_set_error(ctx, GH_ERR_ARG, "test %d of the pattern compares a point with itself", t);
    // continuous steering rotates by any angle: a point of radius r reaches rint(r) on an axis -- the blurred patch of that mode is +-19 px
    for (int e = 0; e < 4; e += 2)
      if ((int)q[e] * q[e] + (int)q[e + 1] * q[e + 1] > 379)  // 19.49^2 = 379.9
        return gh_set_error(ctx, GH_ERR_ARG, "test %d of the pattern has a point beyond radius 19.49 of the keypoint", t);
  }
  for (int k = 0; k < 30; ++k) {
    const double th = (12.0 * k) * (3.14159265358979323846 / 180.0), c = cos(th), s = sin(th);
    for (int t = 0; t < 256; ++t) {
      const int8_t* q = pattern + 4 * t;
      const int v[4] = {rnd(q[0] * c - q[1] * s), rnd(q[0] * s + q[1] * c), rnd(q[2] * c - q[3] * s), rnd(q[2] * s + q[3] * c)};
      for (int e = 0; e < 4; ++e) {
        if (v[e] < -13 || v[e] > 13) {
          if (fits_table) {
            bad_t = t;
            bad_k = k;
          }
          fits_table = false;
        }
        rot[((size_t)k * 256 + t) * 4 + e] = (int8_t)(v[e] < -13 ? -13 : (v[e] > 13 ? 13 : v[e]));
      }
    }
  }
  // The 30-bin table mode blurs +-13 px only: a pattern with points beyond radius 13.49 (the canonical ORB bit_pattern_31_
  // has (-13, -13)) is accepted for continuous steering alone.
  if (!fits_table && p->steer == 0)
    return gh_set_error(ctx, GH_ERR_ARG,
                        "test %d of the pattern leaves the +-13 px blurred patch when rotated by %d degrees (points must lie within "
                        "radius 13.49 of the keypoint for the 30-bin table; gh_orb_plan_set_steering(plan, 1) first takes radius 19.49)",
                        bad_t, 12 * bad_k);
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // extractions in flight still read the old table
  memcpy(p->base_pattern, pattern, sizeof(p->base_pattern));
  p->base_pattern_fits_table = fits_table;
  GH_TRY(gh_dev_upload(ctx, p->d_base_pattern, p->base_pattern, sizeof(p->base_pattern)));
  return upload_pattern(p, rot.data());
}

// Orientation / steering mode of the plan (oracle/orb_oracle.c steps 6 and 8 vs 6' and 8').
//   0  30 orientation bins of 12 degrees, precomputed rotated pattern per bin (the default; integer end to end)
//   1  continuous: angle = the fp32 polynomial arctangent of OpenCV's fastAtan2 (what ORB-SLAM's IC_Angle calls) on the same
//      integer moments, every test point rotated by it and rounded to the nearest pixel (ties to even, cvRound) -- the
//      steering of OpenCV / ORB-SLAM's computeOrbDescriptor, so that with the canonical test pattern installed
//      (gh_orb_plan_set_pattern) the descriptors are those an ORB vocabulary was trained on, up to the image arithmetic
//      (integer pyramid and blur here, float there).  KeyPoint.angle is the continuous angle.
extern "C" gh_status gh_orb_plan_set_steering(gh_orb_plan* p, int mode) {
  if (!p) return GH_ERR_ARG;
  gh_ctx* ctx = p->ctx;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, mode == 0 || mode == 1);
  if (mode == 0 && !p->base_pattern_fits_table)
    return gh_set_error(ctx, GH_ERR_ARG, "the installed test pattern has points beyond radius 13.49: it only works with continuous steering");
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  p->steer = mode;
  // (graphs captured for the other mode hold the other kernel)
  for (auto* v : {&p->graphs, &p->retired})
    for (auto& g : *v) {
      hipEventSynchronize(g.done);
      hipGraphExecDestroy(g.exec);
      hipEventDestroy(g.done);
    }
  p->graphs.clear();
  p->retired.clear();
  return GH_OK;
}

extern "C" gh_status gh_orb_plan_set_distribution(gh_orb_plan* p, int mode) {
  if (!p) return GH_ERR_ARG;
  gh_ctx* ctx = p->ctx;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, mode == 0 || mode == 1);
  if (mode == 1 && !p->qt) {
    GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GH_TRY(gh_qt_create(ctx, p->L, p->lw, p->lh, p->quota, p->max_batch, &p->qt, &p->bytes));
    const char* e = getenv("GSLAM_HIP_QT_PLANE");
    if (!(e && e[0] == '0') && p->pass1 != 0 && p->pk_score) {
      size_t off = 0;
      for (int l = 0; l < p->L; ++l) {
        p->plane_pitch[l] = p->pitch[l] + 128;  // (a multiple of 64: rows start on a 64-byte boundary, like the tile rows)
        p->plane_off[l] = off;
        off += (size_t)p->plane_pitch[l] * p->lh[l];
      }
      p->plane_slab = (off + 255) & ~(size_t)255;
      GH_TRY(plan_alloc(p, (size_t)p->max_batch * p->plane_slab, (void**)&p->score_plane));
      GH_HIP(ctx, hipMemsetAsync(p->score_plane, 0, (size_t)p->max_batch * p->plane_slab, ctx->stream));
      GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
  }
  p->distribution = mode;
  return GH_OK;
}

extern "C" gh_status gh_orb_plan_create(gh_ctx* ctx, int width, int height, int max_batch,
                                        const gh_orb_params* params, gh_orb_plan** out) {
  if (!ctx || !out) return GH_ERR_ARG;
  GH_ENTER(ctx);
  *out = nullptr;
  gh_orb_params prm;
  gh_orb_default_params(&prm);
  if (params) prm = *params;
  GH_CHECK_ARG(ctx, width >= 2 * kEdge + 1 && height >= 2 * kEdge + 1 && width <= 16384 && height <= 16384);
  GH_CHECK_ARG(ctx, max_batch >= 1 && max_batch <= 65535);
  GH_CHECK_ARG(ctx, prm.n_levels >= 1 && prm.n_levels <= kMaxL && prm.n_features >= 1 && prm.n_features <= (1 << 20));
  GH_CHECK_ARG(ctx, prm.min_th_fast >= 1 && prm.ini_th_fast >= prm.min_th_fast && prm.ini_th_fast <= 254);
  GH_HIP(ctx, hipSetDevice(ctx->device));
  gh_orb_plan* p = new (std::nothrow) gh_orb_plan();
  if (!p) return GH_ERR_NOMEM;
  p->ctx = ctx;
  p->w = width;
  p->h = height;
  p->max_batch = max_batch;
  p->prm = prm;
  if (const char* e = getenv("GSLAM_HIP_ORB_FUSE_PYRAMID")) {
    p->fuse_pyramid = atoi(e) == 1;
    p->pyramid_ahead = atoi(e) == 2;
  }
  if (const char* e = getenv("GSLAM_HIP_ORB_PKSCORE")) p->pk_score = atoi(e) != 0;
  if (const char* e = getenv("GSLAM_HIP_ORB_LDSPAD")) p->lds_pad = atoi(e) < 0 ? 0 : atoi(e);
  if (const char* e = getenv("GSLAM_HIP_ORB_DESC_LDSPAD")) p->desc_lds_pad = atoi(e) < 0 ? 0 : atoi(e);
  if (const char* e = getenv("GSLAM_HIP_ORB_RESIZE_MFMA")) p->resize_mfma = atoi(e) != 0;
  if (const char* e = getenv("GSLAM_HIP_ORB_DESC_MFMA")) p->desc_mfma = atoi(e) < 0 ? 0 : (atoi(e) > 2 ? 2 : atoi(e));
  if (const char* e = getenv("GSLAM_HIP_ORB_PASS1")) p->pass1 = atoi(e) != 0;
  const int L = p->L = prm.n_levels;
  // geometry (oracle step 1 / 5): exact integer arithmetic
  long long den = ipow(6, L) - ipow(5, L);
  int qsum = 0;
  size_t off = 0;
  int coff = 0;
  for (int l = 0; l < L; ++l) {
    long long p5 = ipow(5, l), p6 = ipow(6, l);
    p->lw[l] = (int)((2LL * width * p5 + p6) / (2 * p6));
    p->lh[l] = (int)((2LL * height * p5 + p6) / (2 * p6));
    p->pitch[l] = (p->lw[l] + 63) & ~63;
    p->lvl_off[l] = off;
    off += (size_t)p->pitch[l] * p->lh[l];
    off = (off + 255) & ~(size_t)255;
    if (l < L - 1) {
      long long num = (long long)prm.n_features * ipow(5, l) * ipow(6, L - 1 - l);
      p->quota[l] = (int)((2 * num + den) / (2 * den));
      if (p->quota[l] > prm.n_features - qsum) p->quota[l] = prm.n_features - qsum;
      qsum += p->quota[l];
    } else {
      p->quota[l] = prm.n_features - qsum > 0 ? prm.n_features - qsum : 0;
    }
    p->scale[l] = l == 0 ? 1.0f : p->scale[l - 1] * 1.2f;
    const int vw = p->lw[l] - 2 * kEdge, vh = p->lh[l] - 2 * kEdge;
    p->ncx[l] = vw > 0 ? (vw + kCell - 1) / kCell : 0;
    p->ncy[l] = vh > 0 ? (vh + kCell - 1) / kCell : 0;
    if (p->ncx[l] == 0 || p->ncy[l] == 0) p->ncx[l] = p->ncy[l] = 0;
    p->cell_off[l] = coff;
    coff += p->ncx[l] * p->ncy[l];
  }
  // a level without a valid region selects nothing: its quota is simply unused (as in the oracle)
  int qo = 0;
  for (int l = 0; l < L; ++l) {
    p->quota_off[l] = qo;
    qo += p->quota[l];
  }
  p->slab = off + 256;  // tail pad: the tile loader may read a clamped dword at the very end
  p->cells_per_frame = coff > 0 ? coff : 1;
  gh_status st = GH_OK;
  const size_t B = (size_t)max_batch;
  const int K = prm.n_features;
  // resize tables
  size_t tab_words = 0;
  // every table starts 32-byte aligned; x tables are padded to a multiple of 8 entries (pads continue with the next
  // source column, fraction 0, so that the per-thread window bounds hold), y tables to a multiple of 4 (last replicated)
  for (int l = 1; l < L; ++l) tab_words += 3 * (((size_t)p->lw[l] + 7) & ~(size_t)7) + (((size_t)p->lh[l] + 7) & ~(size_t)7);
  // + the MFMA resize's A operands (256 words per 8-column group) and window starts (one word per group, padded to 8)
  for (int l = 1; l < L; ++l) {
    const size_t ngr = ((size_t)p->lw[l] + 7) / 8;
    tab_words += ngr * 256 + ((ngr + 7) & ~(size_t)7);
  }
  // + per source level l < L - 1: ownership of level l + 1 by the tile columns / rows of fast_cells(l), padded to 8 words
  for (int l = 0; l + 1 < L; ++l)
    tab_words += (((size_t)(p->ncx[l] + 1) / 2 + 1 + 7) & ~(size_t)7) + (((size_t)(p->ncy[l] + 1) / 2 + 1 + 7) & ~(size_t)7);
  std::vector<uint32_t> htab(tab_words ? tab_words : 1);
  do {
    if ((st = plan_alloc(p, B * p->slab, (void**)&p->pyr)) != GH_OK) break;
    if ((st = plan_alloc(p, B * p->cells_per_frame * kCellRec * sizeof(uint32_t), (void**)&p->cell_cnt)) != GH_OK) break;
    if ((st = plan_alloc(p, B * p->cells_per_frame * kCap * sizeof(uint32_t), (void**)&p->cell_ent)) != GH_OK) break;
    if ((st = plan_alloc(p, B * K * sizeof(SelKp), (void**)&p->sel)) != GH_OK) break;
    if ((st = plan_alloc(p, B * kMaxL * sizeof(int32_t), (void**)&p->level_cnt)) != GH_OK) break;
    if ((st = plan_alloc(p, sizeof(GH_ORB_PATTERN), (void**)&p->d_pattern)) != GH_OK) break;
    if ((st = plan_alloc(p, 256 * 4, (void**)&p->d_base_pattern)) != GH_OK) break;
    if ((st = plan_alloc(p, sizeof(GH_ORB_DIR), (void**)&p->d_dir)) != GH_OK) break;
    if ((st = plan_alloc(p, htab.size() * sizeof(uint32_t), (void**)&p->tabs)) != GH_OK) break;
    size_t tw = 0;
    for (int l = 1; l < L; ++l) {
      for (int axis = 0; axis < 2; ++axis) {
        const int n_src = axis == 0 ? p->lw[l - 1] : p->lh[l - 1];
        const int n_dst = axis == 0 ? p->lw[l] : p->lh[l];
        (axis == 0 ? p->xtab[l] : p->ytab[l]) = p->tabs + tw;
        for (int x = 0; x < n_dst; ++x) {
          long long P = ((long long)(2 * x + 1) * n_src * 2048) / (2LL * n_dst) - 1024;
          if (P < 0) P = 0;
          int sx = (int)(P >> 11), fx = (int)(P & 2047);
          if (sx >= n_src - 1) {
            sx = n_src - 1;
            fx = 0;
          }
          htab[tw++] = ((uint32_t)sx << 16) | (uint32_t)fx;
        }
        while (tw & 7) {  // keeps every table 32-byte aligned
          htab[tw] = axis == 0 ? ((htab[tw - 1] >> 16) + 1u) << 16 : htab[tw - 1];
          ++tw;
        }
        if (axis == 0) {
          // window contract of resize_kernel: within a group of 8 columns, taps of columns 0..3 lie in window bytes
          // 0..7 and taps of columns 4..7 in bytes 4..11 (true for the 1.2 scale; refuse anything else loudly)
          const size_t x0 = (size_t)(p->xtab[l] - p->tabs), nx_pad = tw - x0;
          const uint32_t* t = htab.data() + x0;
          for (size_t g = 0; g + 8 <= nx_pad; g += 8)
            for (int i = 0; i < 8; ++i) {
              const int o = (int)(t[g + i] >> 16) - (int)(t[g] >> 16) - (i >= 4 ? 4 : 0);
              if (o < 0 || o + 1 > 7) st = GH_ERR_ARG;
            }
          // the per-column selector / weight tables derived from it (ResizeTabs)
          p->xsel[l] = p->tabs + tw;
          p->xwgt[l] = p->tabs + tw + nx_pad;
          for (size_t c = 0; c < nx_pad; ++c) {
            const int sx = (int)(t[c] >> 16), sx0 = (int)(t[c & ~(size_t)7] >> 16), hi = (c & 7) >= 4 ? 4 : 0;
            const int o = sx - sx0 - hi;
            int sx1 = sx + 1 < n_src ? sx + 1 : n_src - 1;
            if (sx1 < sx) sx1 = sx;  // pad entries past the last source column
            const int o1 = sx1 - sx0 - hi;
            const uint32_t fx = t[c] & 0xFFFFu;
            htab[tw + c] = 0x0c000c00u | ((uint32_t)(o1 & 7) << 16) | (uint32_t)(o & 7);
            htab[tw + nx_pad + c] = (fx << 16) | (2048u - fx);
          }
          tw += 2 * nx_pad;
        }
      }
    }
    if (st != GH_OK) {
      gh_set_error(ctx, st, "resize table violates the 8-column window contract (level size ratio is not ~1.2)");
      break;
    }
    // MFMA resize tables of level l (as a destination): see ResizeTabs / resize_tile_mfma
    for (int l = 1; l < L; ++l) {
      const uint32_t* xt = htab.data() + (p->xtab[l] - p->tabs);
      const int wd = p->lw[l], ngr = (wd + 7) / 8, n_src = p->lw[l - 1];
      auto f16_of = [](int v) -> uint32_t {  // 0 <= v <= 2048: exact
        if (v == 0) return 0u;
        int e = 0;
        while ((v >> (e + 1)) != 0) ++e;
        const int mant = e <= 10 ? (v << (10 - e)) : (v >> (e - 10));  // (v = 2048 has e = 11: a shift by -1 is undefined)
        return (uint32_t)(((e + 15) << 10) | (mant & 0x3FF));
      };
      uint32_t* mt = htab.data() + tw;
      uint32_t* mc = htab.data() + tw + (size_t)ngr * 256;
      bool ok = p->resize_mfma;
      for (int g = 0; g < ngr; ++g) {
        const int cw = (int)(xt[8 * g] >> 16) & ~7;
        mc[g] = (uint32_t)cw;
        for (int lane = 0; lane < 64; ++lane) {
          const int m = lane & 15, x = 8 * g + m;
          uint32_t wds[4] = {0, 0, 0, 0};
          if (x < wd) {
            const int sx = (int)(xt[x] >> 16), fx = (int)(xt[x] & 0xFFFFu);
            const int k0 = sx - cw, k1 = (sx + 1 < n_src ? sx + 1 : n_src - 1) - cw;
            if (k0 < 0 || k1 > 31) ok = false;  // the 32-column K window must hold both taps of all 16 columns
            for (int e = 0; e < 8; ++e) {
              const int k = 8 * (lane >> 4) + e;
              int wgt = 0;
              if (k == k0) wgt += 2048 - fx;
              if (k == k1) wgt += fx;  // (k1 == k0 at the right border, where fx = 0)
              wds[e >> 1] |= f16_of(wgt) << (16 * (e & 1));
            }
          }
          for (int e = 0; e < 4; ++e) mt[((size_t)g * 64 + lane) * 4 + e] = wds[e];
        }
      }
      p->mtab[l] = ok ? p->tabs + tw : nullptr;
      p->mcw[l] = ok ? p->tabs + tw + (size_t)ngr * 256 : nullptr;
      tw += (size_t)ngr * 256 + (((size_t)ngr + 7) & ~(size_t)7);
    }
    // ownership of level l + 1 inside fast_cells(l): tile column bx (level-l columns 64 bx .. 64 bx + 63 plus halo) owns
    // the output 8-groups whose first source column lies in [64 bx, 64 bx + 64); tile row by (rows from 64 by + 15) the
    // output rows whose source row lies in [64 by + 15, 64 by + 79); the first / last tile row and the last tile column
    // also take what lies outside every tile.  Any partition is CORRECT (resize_item reads global memory); this one
    // makes the reads hit the lines the tile load has just fetched.
    for (int l = 0; l + 1 < L; ++l) {
      const int nbx = (p->ncx[l] + 1) / 2, nby = (p->ncy[l] + 1) / 2;
      const uint32_t* xt = htab.data() + (p->xtab[l + 1] - p->tabs);
      const uint32_t* yt = htab.data() + (p->ytab[l + 1] - p->tabs);
      const int ngroups = (p->lw[l + 1] + 7) / 8, hd = p->lh[l + 1];
      p->own_gx[l] = reinterpret_cast<const int32_t*>(p->tabs + tw);
      int g = 0;
      for (int bx = 0; bx <= nbx; ++bx) {
        if (bx == nbx) g = ngroups;
        else
          while (bx > 0 && g < ngroups && (int)(xt[8 * g] >> 16) < 64 * bx) ++g;
        htab[tw++] = (uint32_t)g;
      }
      while (tw & 7) htab[tw++] = (uint32_t)ngroups;
      p->own_gy[l] = reinterpret_cast<const int32_t*>(p->tabs + tw);
      int y = 0;
      for (int by = 0; by <= nby; ++by) {
        if (by == nby) y = hd;
        else
          while (by > 0 && y < hd && (int)(yt[y] >> 16) < 64 * by + 15) ++y;
        htab[tw++] = (uint32_t)y;
      }
      while (tw & 7) htab[tw++] = (uint32_t)hd;
    }
    if ((st = gh_dev_upload(ctx, p->tabs, htab.data(), htab.size() * sizeof(uint32_t))) != GH_OK) break;
    if ((st = upload_pattern(p, &GH_ORB_PATTERN[0][0][0])) != GH_OK) break;
    memcpy(p->base_pattern, &GH_ORB_PATTERN[0][0][0], sizeof(p->base_pattern));  // bin 0 = the unrotated tests
    if ((st = gh_dev_upload(ctx, p->d_base_pattern, p->base_pattern, sizeof(p->base_pattern))) != GH_OK) break;
    if ((st = gh_dev_upload(ctx, p->d_dir, GH_ORB_DIR, sizeof(GH_ORB_DIR))) != GH_OK) break;
  } while (0);
  if (st != GH_OK) {
    gh_orb_plan_destroy(p);
    return st;
  }
  *out = p;
  return GH_OK;
}

extern "C" gh_status gh_orb_plan_level(const gh_orb_plan* p, int level, int* w, int* h, int* quota) {
  if (!p || level < 0 || level >= p->L) return GH_ERR_ARG;
  if (w) *w = p->lw[level];
  if (h) *h = p->lh[level];
  if (quota) *quota = p->quota[level];
  return GH_OK;
}

extern "C" size_t gh_orb_plan_device_bytes(const gh_orb_plan* p) { return p ? p->bytes : 0; }

static gh_status orb_enqueue(gh_orb_plan* p, const uint8_t* gray_dev, int batch, size_t frame_stride, int row_stride,
                             gh_keypoint* kps_dev, uint8_t* desc_dev, int32_t* counts_dev);

extern "C" gh_status gh_orb_extract_dev(gh_orb_plan* p, const uint8_t* gray_dev, int batch, size_t frame_stride,
                                        int row_stride, gh_keypoint* kps_dev, uint8_t* desc_dev,
                                        int32_t* counts_dev) {
  if (!p) return GH_ERR_ARG;
  gh_ctx* ctx = p->ctx;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, batch >= 0 && batch <= p->max_batch);
  if (batch == 0) return GH_OK;
  GH_CHECK_ARG(ctx, gray_dev && kps_dev && desc_dev && counts_dev && row_stride >= p->w && row_stride < (1 << 24));  // (24-bit row offsets)
  // (32-bit byte offsets inside a level: a view into a very wide buffer -- an ROI -- can pass the test above and still wrap)
  GH_CHECK_ARG(ctx, (uint64_t)row_stride * (uint64_t)p->h < (1ull << 32));
  GH_CHECK_ARG(ctx, frame_stride >= (size_t)row_stride * p->h || batch == 1);
  GH_CHECK_ARG(ctx, ((uintptr_t)desc_dev & 7) == 0 && ((uintptr_t)kps_dev & 3) == 0);
  // GSLAM_HIP_ORB_GRAPH=0: always launch kernel by kernel (A/B measurements)
  static const bool graph_env = [] {
    const char* e = getenv("GSLAM_HIP_ORB_GRAPH");
    return !(e && e[0] == '0');
  }();
  const bool small = (long long)batch * p->w * p->h <= (4LL << 20);  // up to two 1080p frames: launch-bound
  if (p->distribution != 0) {
    // quadtree mode: when the plan's key budget cut a candidate list below its worst case, a list that overflowed is an error of
    // THIS call -- it waits for its own kernels.  Otherwise (every plan whose worst case fits 4 GB: 1080p up to ~400 frames per
    // call) nothing can overflow and the call is asynchronous like the default mode's.
    GH_TRY(orb_enqueue(p, gray_dev, batch, frame_stride, row_stride, kps_dev, desc_dev, counts_dev));
    if (!gh_qt_can_overflow(p->qt)) return GH_OK;
    GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return gh_qt_check(ctx, p->qt);
  }
  if (!(graph_env && small && !p->graphs_off && !ctx->prof_on && !p->dbg_on && ctx->stream != nullptr))
    return orb_enqueue(p, gray_dev, batch, frame_stride, row_stride, kps_dev, desc_dev, counts_dev);
  for (auto& g : p->graphs)
    if (g.gray == gray_dev && g.kps == kps_dev && g.desc == desc_dev && g.counts == counts_dev && g.batch == batch &&
        g.row_stride == row_stride && g.frame_stride == frame_stride) {
      ++p->graph_hits;
      GH_HIP(ctx, hipGraphLaunch(g.exec, ctx->stream));
      GH_HIP(ctx, hipEventRecord(g.done, ctx->stream));
      return GH_OK;
    }
  // retired graphs whose last launch has completed can go now
  for (size_t k = 0; k < p->retired.size();) {
    if (hipEventQuery(p->retired[k].done) == hipSuccess) {
      hipGraphExecDestroy(p->retired[k].exec);
      hipEventDestroy(p->retired[k].done);
      p->retired.erase(p->retired.begin() + (long)k);
    } else {
      (void)hipGetLastError();
      ++k;
    }
  }
  ++p->graph_misses;
  if (p->graph_misses >= gh_orb_plan::kGraphGiveUp && p->graph_misses > 4 * p->graph_hits) {
    p->graphs_off = true;  // the caller's argument sets do not repeat: capturing costs more than it saves
    return orb_enqueue(p, gray_dev, batch, frame_stride, row_stride, kps_dev, desc_dev, counts_dev);
  }
  // first call with these arguments: capture the launch sequence on the plan's own stream (nothing executes during the
  // capture, and nobody else can enqueue there), then replay it on the caller's
  if (!p->cap_stream && hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    p->cap_stream = nullptr;
    p->graphs_off = true;
    return orb_enqueue(p, gray_dev, batch, frame_stride, row_stride, kps_dev, desc_dev, counts_dev);
  }
  if (hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
    (void)hipGetLastError();
    p->graphs_off = true;
    return orb_enqueue(p, gray_dev, batch, frame_stride, row_stride, kps_dev, desc_dev, counts_dev);
  }
  hipStream_t const user_stream = ctx->stream;
  ctx->stream = p->cap_stream;
  p->capturing = true;
  const gh_status st = orb_enqueue(p, gray_dev, batch, frame_stride, row_stride, kps_dev, desc_dev, counts_dev);
  p->capturing = false;
  ctx->stream = user_stream;
  hipGraph_t graph = nullptr;
  const hipError_t ee = hipStreamEndCapture(p->cap_stream, &graph);
  hipGraphExec_t exec = nullptr;
  hipEvent_t done = nullptr;
  if (st != GH_OK || ee != hipSuccess || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess ||
      hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    if (exec) hipGraphExecDestroy(exec);
    if (graph) hipGraphDestroy(graph);
    p->graphs_off = true;
    if (st != GH_OK) return st;
    return orb_enqueue(p, gray_dev, batch, frame_stride, row_stride, kps_dev, desc_dev, counts_dev);
  }
  hipGraphDestroy(graph);
  if ((int)p->graphs.size() >= gh_orb_plan::kGraphRing) {  // the oldest set goes; its exec may still be queued on the stream
    p->retired.push_back(p->graphs.front());
    p->graphs.erase(p->graphs.begin());
  }
  p->graphs.push_back({gray_dev, kps_dev, desc_dev, counts_dev, batch, row_stride, frame_stride, exec, done});
  GH_HIP(ctx, hipGraphLaunch(exec, ctx->stream));
  GH_HIP(ctx, hipEventRecord(done, ctx->stream));
  return GH_OK;
}

static gh_status orb_enqueue(gh_orb_plan* p, const uint8_t* gray_dev, int batch, size_t frame_stride, int row_stride,
                             gh_keypoint* kps_dev, uint8_t* desc_dev, int32_t* counts_dev) {
  gh_ctx* ctx = p->ctx;
  const int L = p->L, K = p->prm.n_features;

  LevelView lv[kMaxL];
  // zero-copy level 0 reads whole 16-byte windows of every padded row, the last row included, so it needs the full
  // row_stride * h bytes of every frame to be readable; a single frame handed over with a smaller frame_stride (an ROI
  // view whose allocation ends at (h-1) * row_stride + w) is staged through the plan's own level-0 slab instead
  const bool aligned0 = ((uintptr_t)gray_dev & 15) == 0 && (row_stride & 15) == 0 && (frame_stride & 15) == 0 &&
                        frame_stride >= (size_t)row_stride * p->h;
  if (aligned0) {
    lv[0] = {gray_dev, frame_stride, row_stride, p->w, p->h};
  } else {
    dim3 grid(gh_div_up(p->w, 256), p->h, batch);
    GH_LAUNCH(ctx, "orb_copy_level0", copy_rows_kernel, grid, dim3(256), 0, gray_dev, frame_stride, row_stride,
              p->pyr + p->lvl_off[0], p->slab, p->pitch[0], p->w, p->h);
    lv[0] = {p->pyr + p->lvl_off[0], p->slab, p->pitch[0], p->w, p->h};
  }
  for (int l = 1; l < L; ++l) lv[l] = {p->pyr + p->lvl_off[l], p->slab, p->pitch[l], p->lw[l], p->lh[l]};
  uint32_t* dbg = p->dbg_on ? p->dbg : nullptr;

  // Level l + 1 is produced inside fast_cells(l) (see the kernel); a level whose predecessor runs no FAST pass (no valid
  // region or no quota) is produced by the stand-alone resize launch instead.
  auto resize_standalone = [&](int l) -> gh_status {
    const int gpr = gh_div_up(p->lw[l], kResizeCols), nrg = gh_div_up(p->lh[l], kResizeRows);
    const int n_items = gpr * nrg;
    const uint32_t inv = (uint32_t)((0x100000000ull + (uint64_t)gpr - 1) / (uint64_t)gpr);  // exact for item < 2^32 / gpr
    GH_CHECK_ARG(ctx, (uint64_t)n_items * (uint64_t)gpr < 0x100000000ull);
    // only the caller's own level-0 buffer can end right after its last row
    const int unsafe_frame = (l == 1 && aligned0) ? batch - 1 : -1;
    GH_LAUNCH(ctx, "orb_resize", resize_kernel, dim3(gh_div_up(n_items, 256), batch), dim3(256), 0, lv[l - 1],
              p->pyr + p->lvl_off[l], p->slab, p->pitch[l], p->lw[l], p->lh[l],
              ResizeTabs{p->xtab[l], p->xsel[l], p->xwgt[l], p->ytab[l]}, gpr, inv, n_items, unsafe_frame);
    return GH_OK;
  };
  SelectArgs sa;
  for (int l = 0; l < kMaxL; ++l) {
    sa.ncells[l] = l < L ? p->ncx[l] * p->ncy[l] : 0;
    sa.cell_off[l] = l < L ? p->cell_off[l] : 0;
    sa.ncx[l] = l < L && p->ncx[l] > 0 ? p->ncx[l] : 1;
    sa.quota[l] = l < L ? p->quota[l] : 0;
    sa.quota_off[l] = l < L ? p->quota_off[l] : 0;
  }
  int max_cells = 0;
  for (int l = 0; l < L; ++l) max_cells = sa.ncells[l] > max_cells ? sa.ncells[l] : max_cells;
  static const bool no_cache = [] {
    const char* e = getenv("GSLAM_HIP_ORB_SELECT_CACHED");  // "0": stream the records (A/B measurements)
    return e && e[0] == '0';
  }();
  const bool cached = max_cells <= kSelCached * 256 && !no_cache;
  auto launch_select = [&](int l0, int nl) -> gh_status {
    if (cached)
      GH_LAUNCH(ctx, "orb_select", select_kernel<true>, dim3(nl, batch), dim3(256), 0, sa, p->cell_cnt, p->cell_ent,
                p->cells_per_frame, K, p->sel, p->level_cnt, dbg, l0);
    else
      GH_LAUNCH(ctx, "orb_select", select_kernel<false>, dim3(nl, batch), dim3(256), 0, sa, p->cell_cnt, p->cell_ent,
                p->cells_per_frame, K, p->sel, p->level_cnt, dbg, l0);
    return GH_OK;
  };
  // select(l) needs fast_cells(l) only, and it is a latency-bound kernel (histogram + three passes over the cell
  // records): in a batched call it runs on a side stream beside the VALU-bound fast_cells of the levels that follow,
  // instead of as one launch behind the last level.  Small calls (below 16 Mpixel) keep the one launch (8 more launches and
  // 9 event operations would cost more than the overlap gives).  GSLAM_HIP_ORB_SELECT_OVERLAP=0 / 1 forces either.
  static const int overlap_env = [] {
    const char* e = getenv("GSLAM_HIP_ORB_SELECT_OVERLAP");
    return e ? (e[0] == '0' ? 0 : 1) : -1;
  }();
  bool overlap = overlap_env < 0 ? (long long)batch * p->w * p->h >= (16LL << 20) : overlap_env == 1;  // >= 8 frames of 1080p
  // (quadtree mode from the score plane: the cells of level l run on the same side stream beside the tile kernel of level l + 1)
  // -- measured: 2.30 ms per 100 x 1080p against 2.23 on one stream (the two kernels do not overlap on this part, the events
  // cost); kept behind GSLAM_HIP_QT_SIDE=1
  static const bool qt_side_env = [] { const char* e = getenv("GSLAM_HIP_QT_SIDE"); return e && e[0] == '1'; }();
  bool qt_side = qt_side_env && p->distribution != 0 && p->score_plane != nullptr && !p->capturing && overlap;
  if (p->capturing || p->distribution != 0) overlap = false;  // (a captured call is a small one: one select launch)
  if ((overlap || qt_side) && !p->side) {
    if (hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) != hipSuccess) {
      p->side = nullptr;
      overlap = qt_side = false;
    } else {
      bool ok = hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming) == hipSuccess;
      for (int l = 0; l < kMaxL && ok; ++l) ok = hipEventCreateWithFlags(&p->ev_level[l], hipEventDisableTiming) == hipSuccess;
      if (!ok) return gh_set_error(ctx, GH_ERR_HIP, "gh_orb_extract_dev: event creation failed");
    }
  }
  struct StreamSwap {  // GH_LAUNCH launches (and profiles) on ctx->stream
    gh_ctx* c;
    hipStream_t keep;
    StreamSwap(gh_ctx* c_, hipStream_t s) : c(c_), keep(c_->stream) { c->stream = s; }
    ~StreamSwap() { c->stream = keep; }
  };
  // small calls: pyramid first, then every level in one FAST launch (GSLAM_HIP_ORB_ALL_LEVELS=0: A/B measurements)
  static const bool all_env = [] {
    const char* e = getenv("GSLAM_HIP_ORB_ALL_LEVELS");
    return !(e && e[0] == '0');
  }();
  const bool quadtree = p->distribution != 0;
  const bool all_levels = (all_env && (long long)batch * p->w * p->h <= (4LL << 20)) || quadtree;
  if (quadtree) {
    // ORB-SLAM's distribution (oracle steps 4', 5'): the whole pyramid first, then cells + tree of orb_quadtree.hip leave sel /
    // level_cnt as orb_select would
    LevelView planes[kMaxL];
    for (int l = 0; l < kMaxL; ++l) planes[l] = LevelView{nullptr, 0, 0, 0, 0};
    if (p->score_plane != nullptr) {
      // S of every level by the default mode's tile kernel (plane variant: no cell stage), the next level fused as there; a
      // level whose cells do not fit the wave-per-cell kernel keeps the image path of orb_quadtree.hip.  The cells of level l
      // (latency bound: LDS round trips per cell) run on the side stream beside the tile kernel of level l + 1 (VALU bound).
      const bool side_ok = qt_side && p->side != nullptr && !dbg;
      GH_TRY(gh_qt_begin(ctx, p->qt, batch));
      auto cells_of = [&](int l, const LevelView* plane) -> gh_status {
        if (!side_ok) return gh_qt_cells(ctx, p->qt, l, lv[l], plane, batch, p->prm.min_th_fast, p->prm.ini_th_fast);
        GH_HIP(ctx, hipEventRecord(p->ev_level[l], ctx->stream));  // level l's image / plane (and the cleared counters) exist
        GH_HIP(ctx, hipStreamWaitEvent(p->side, p->ev_level[l], 0));
        StreamSwap sw(ctx, p->side);
        return gh_qt_cells(ctx, p->qt, l, lv[l], plane, batch, p->prm.min_th_fast, p->prm.ini_th_fast);
      };
      for (int l = 0; l < L; ++l) {
        const bool plane = p->ncx[l] != 0 && gh_qt_plane_ok(p->qt, l);
        if (!plane) {
          GH_TRY(cells_of(l, nullptr));
          if (l + 1 < L) GH_TRY(resize_standalone(l + 1));
          continue;
        }
        NextLevel nx{nullptr, 0, 0, 0, ResizeTabs{nullptr, nullptr, nullptr, nullptr}, nullptr, nullptr, -1};
        if (l + 1 < L && p->fuse_pyramid)
          nx = NextLevel{p->pyr + p->lvl_off[l + 1], p->slab, p->pitch[l + 1], p->lh[l + 1],
                         ResizeTabs{p->xtab[l + 1], p->xsel[l + 1], p->xwgt[l + 1], p->ytab[l + 1], p->mtab[l + 1], p->mcw[l + 1]},
                         p->own_gx[l], p->own_gy[l], (l == 0 && aligned0) ? batch - 1 : -1};
        nx.plane = p->score_plane + p->plane_off[l];
        nx.plane_frame_stride = p->plane_slab;
        nx.plane_pitch = p->plane_pitch[l];
        const long long tiles = (long long)gh_div_up(p->ncx[l], 2) * gh_div_up(p->ncy[l], 2) * batch;
        GH_CHECK_ARG(ctx, tiles < (1LL << 30));
        const uint32_t nbx = (uint32_t)gh_div_up(p->ncx[l], 2), tpf = nbx * (uint32_t)gh_div_up(p->ncy[l], 2);
        nx.tiles_inv = magic_div(tpf, (uint32_t)tiles);
        nx.nbx_inv = magic_div(nbx, tpf);
        // GSLAM_HIP_ORB_PLANE_SW=1: the plane by the barrier-free sliding-window kernel (round-6 experiment), the next level by the
        // stand-alone resize launch
        const char* sw_env = getenv("GSLAM_HIP_ORB_PLANE_SW");  // (read per call: the experiment's A/B runs switch inside one process)
        const int plane_sw = sw_env ? atoi(sw_env) : 0;  // bit 0: on; bits 1-3: A/B variants of the kernel
        if (plane_sw & 1) {
          const int nstrips = gh_div_up(p->lw[l] - 3, kSwOwn), nchunks = gh_div_up(p->lh[l] - kEdge, kSwRows);
          const long long waves = (long long)nstrips * nchunks * batch;
          GH_CHECK_ARG(ctx, waves < (1LL << 30));
          GH_LAUNCH(ctx, "orb_fast_plane_sw", fast_plane_sw_kernel, dim3((unsigned)gh_div_up(waves, 4)), dim3(256), 0, lv[l],
                    p->prm.min_th_fast, nstrips, nchunks, batch, nx.plane, nx.plane_frame_stride, nx.plane_pitch, plane_sw);
          if (l + 1 < L) GH_TRY(resize_standalone(l + 1));
        } else {
        GH_LAUNCH(ctx, "orb_fast_plane", (fast_cells_kernel<true, 1, true>), dim3(8 * gh_div_up(tiles, 8)), dim3(256), p->lds_pad, lv[l],
                  p->ncx[l], p->ncy[l], p->prm.min_th_fast, p->prm.ini_th_fast, p->cell_cnt, p->cell_ent, p->cells_per_frame,
                  p->cell_off[l], batch, nx, dbg);
        if (l + 1 < L && !p->fuse_pyramid) GH_TRY(resize_standalone(l + 1));
        }
        planes[l] = LevelView{p->score_plane + p->plane_off[l], p->plane_slab, p->plane_pitch[l], p->lw[l], p->lh[l]};
        GH_TRY(cells_of(l, &planes[l]));
      }
      if (side_ok) {
        GH_HIP(ctx, hipEventRecord(p->ev_join, p->side));
        GH_HIP(ctx, hipStreamWaitEvent(ctx->stream, p->ev_join, 0));
      }
      GH_TRY(gh_qt_tree(ctx, p->qt, batch, p->quota_off, K, p->sel, p->level_cnt));
    } else {
      for (int l = 1; l < L; ++l) GH_TRY(resize_standalone(l));
      GH_TRY(gh_qt_enqueue(ctx, p->qt, lv, batch, p->prm.min_th_fast, p->prm.ini_th_fast, p->quota_off, K, p->sel, p->level_cnt, planes));
    }
    overlap = false;
  } else if (all_levels) {
    for (int l = 1; l < L; ++l) GH_TRY(resize_standalone(l));
    AllLevels A;
    A.n_levels = L;
    int tiles = 0;
    for (int l = 0; l < kMaxL; ++l) {
      A.tile_start[l] = tiles;
      A.lv[l] = lv[l < L ? l : 0];
      A.ncx[l] = l < L ? p->ncx[l] : 0;
      A.ncy[l] = l < L ? p->ncy[l] : 0;
      A.cell_off[l] = l < L ? p->cell_off[l] : 0;
      if (l < L && p->ncx[l] != 0 && p->quota[l] > 0) tiles += gh_div_up(p->ncx[l], 2) * gh_div_up(p->ncy[l], 2) * batch;
    }
    A.tile_start[kMaxL] = tiles;
    if (tiles > 0) {
#define GH_FAST_ALL(PK_, P1_)                                                                                            \
  GH_LAUNCH(ctx, "orb_fast_cells", (fast_cells_all_kernel<PK_, P1_>), dim3(tiles), dim3(256), 0, A, p->prm.min_th_fast, \
            p->prm.ini_th_fast, p->cell_cnt, p->cell_ent, p->cells_per_frame, batch, dbg)
      if (!p->pk_score) GH_FAST_ALL(false, 0);
      else if (p->pass1 == 0) GH_FAST_ALL(true, 0);
      else GH_FAST_ALL(true, 1);
#undef GH_FAST_ALL
    }
    overlap = false;
  }
  // pyramid ahead: levels 1 .. L-1 by the stand-alone resize kernel on a stream of their own, each FAST pass waits for its
  // level only -- the memory-instruction-bound resize chain shares the CUs with the VALU-bound FAST passes of earlier levels
  bool ahead = p->pyramid_ahead && !all_levels && !p->capturing && L > 1;
  if (ahead && !p->pyr_stream) {
    bool ok = hipStreamCreateWithFlags(&p->pyr_stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&p->ev_pyr_start, hipEventDisableTiming) == hipSuccess;
    for (int l = 0; l < kMaxL && ok; ++l) ok = hipEventCreateWithFlags(&p->ev_pyr[l], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      ahead = false;
    }
  }
  if (ahead) {
    GH_HIP(ctx, hipEventRecord(p->ev_pyr_start, ctx->stream));  // level 0 (the caller's frames, or the staging copy) is ready
    GH_HIP(ctx, hipStreamWaitEvent(p->pyr_stream, p->ev_pyr_start, 0));
    StreamSwap sw(ctx, p->pyr_stream);
    for (int l = 1; l < L; ++l) {
      GH_TRY(resize_standalone(l));
      GH_HIP(ctx, hipEventRecord(p->ev_pyr[l], p->pyr_stream));
    }
  }
  for (int l = 0; l < L && !all_levels; ++l) {
    const bool fast = p->ncx[l] != 0 && p->quota[l] > 0;
    if (ahead) {
      if (l > 0) GH_HIP(ctx, hipStreamWaitEvent(ctx->stream, p->ev_pyr[l], 0));
      if (fast) {
        const NextLevel none{nullptr, 0, 0, 0, ResizeTabs{nullptr, nullptr, nullptr, nullptr}, nullptr, nullptr, -1};
        const NextLevel& nx = none;
        const long long tiles = (long long)gh_div_up(p->ncx[l], 2) * gh_div_up(p->ncy[l], 2) * batch;
        GH_CHECK_ARG(ctx, tiles < (1LL << 30));
        dim3 grid(8 * gh_div_up(tiles, 8));
        if (!p->pk_score) GH_LAUNCH(ctx, "orb_fast_cells", (fast_cells_kernel<false, 0>), grid, dim3(256), p->lds_pad, lv[l], p->ncx[l], p->ncy[l], p->prm.min_th_fast, p->prm.ini_th_fast, p->cell_cnt, p->cell_ent, p->cells_per_frame, p->cell_off[l], batch, nx, dbg);
        else if (p->pass1 == 0) GH_LAUNCH(ctx, "orb_fast_cells", (fast_cells_kernel<true, 0>), grid, dim3(256), p->lds_pad, lv[l], p->ncx[l], p->ncy[l], p->prm.min_th_fast, p->prm.ini_th_fast, p->cell_cnt, p->cell_ent, p->cells_per_frame, p->cell_off[l], batch, nx, dbg);
        else GH_LAUNCH(ctx, "orb_fast_cells", (fast_cells_kernel<true, 1>), grid, dim3(256), p->lds_pad, lv[l], p->ncx[l], p->ncy[l], p->prm.min_th_fast, p->prm.ini_th_fast, p->cell_cnt, p->cell_ent, p->cells_per_frame, p->cell_off[l], batch, nx, dbg);
      }
    } else if (!fast) {
      if (l + 1 < L) GH_TRY(resize_standalone(l + 1));
    } else {
      NextLevel nx{nullptr, 0, 0, 0, ResizeTabs{nullptr, nullptr, nullptr, nullptr}, nullptr, nullptr, -1};
      if (l + 1 < L && p->fuse_pyramid)
        nx = NextLevel{p->pyr + p->lvl_off[l + 1], p->slab, p->pitch[l + 1], p->lh[l + 1],
                       ResizeTabs{p->xtab[l + 1], p->xsel[l + 1], p->xwgt[l + 1], p->ytab[l + 1], p->mtab[l + 1], p->mcw[l + 1]},
                       p->own_gx[l], p->own_gy[l], (l == 0 && aligned0) ? batch - 1 : -1};
      const long long tiles = (long long)gh_div_up(p->ncx[l], 2) * gh_div_up(p->ncy[l], 2) * batch;
      GH_CHECK_ARG(ctx, tiles < (1LL << 30));
      dim3 grid(8 * gh_div_up(tiles, 8));
      {
        const uint32_t nbx = (uint32_t)gh_div_up(p->ncx[l], 2), tpf = nbx * (uint32_t)gh_div_up(p->ncy[l], 2);
        nx.tiles_inv = magic_div(tpf, (uint32_t)tiles);
        nx.nbx_inv = magic_div(nbx, tpf);
      }
#define GH_FAST(PK_, P1_)                                                                                                     \
  GH_LAUNCH(ctx, "orb_fast_cells", (fast_cells_kernel<PK_, P1_>), grid, dim3(256), p->lds_pad, lv[l], p->ncx[l], p->ncy[l],            \
            p->prm.min_th_fast, p->prm.ini_th_fast, p->cell_cnt, p->cell_ent, p->cells_per_frame, p->cell_off[l], batch, nx, \
            dbg)
      if (!p->pk_score) GH_FAST(false, 0);
      else if (p->pass1 == 0) GH_FAST(true, 0);
      else GH_FAST(true, 1);
#undef GH_FAST
      if (l + 1 < L && !p->fuse_pyramid) GH_TRY(resize_standalone(l + 1));
    }
    if (overlap) {  // (a level without a FAST pass still gets its level_cnt = 0 from select)
      GH_HIP(ctx, hipEventRecord(p->ev_level[l], ctx->stream));
      GH_HIP(ctx, hipStreamWaitEvent(p->side, p->ev_level[l], 0));
      StreamSwap sw(ctx, p->side);
      GH_TRY(launch_select(l, 1));
    }
  }
  if (overlap) {
    GH_HIP(ctx, hipEventRecord(p->ev_join, p->side));
    GH_HIP(ctx, hipStreamWaitEvent(ctx->stream, p->ev_join, 0));
  } else if (!quadtree) {
    GH_TRY(launch_select(0, L));
  }
  if (!cached && dbg) GH_HIP(ctx, hipMemsetAsync(dbg + kDbgSelStreamed, 1, 1, ctx->stream));
  {
    DescribeArgs a;
    for (int l = 0; l < kMaxL; ++l) {
      a.lv[l] = l < L ? lv[l] : lv[0];
      a.quota[l] = l < L ? p->quota[l] : 0;
      a.quota_off[l] = l < L ? p->quota_off[l] : 0;
      a.scale[l] = l < L ? p->scale[l] : 1.0f;
    }
    a.nlevels = L;
    a.blk_off[0] = 0;
    for (int l = 0; l < kMaxL; ++l) a.blk_off[l + 1] = a.blk_off[l] + (l < L ? gh_div_up(a.quota[l], 4 * kDescPipe) : 0);
    DevTables tb{p->d_pattern, p->d_dir, p->d_base_pattern};
    const long long blocks = (long long)gh_div_up(K, 4) * batch;
    GH_CHECK_ARG(ctx, blocks < (1LL << 30));
    if (p->steer == 0 && p->desc_mfma == 2) {
      const long long pblocks = (long long)a.blk_off[L] * batch;
      GH_LAUNCH(ctx, "orb_describe", describe_pipe_kernel, dim3(8 * gh_div_up(pblocks, 8)), dim3(256), p->desc_lds_pad, a, tb, K,
                p->sel, p->level_cnt, kps_dev, desc_dev, counts_dev, batch, dbg);
    } else if (p->steer == 0 && p->desc_mfma)
      GH_LAUNCH(ctx, "orb_describe", (describe_kernel<13, false, true>), dim3(8 * gh_div_up(blocks, 8)), dim3(256), p->desc_lds_pad, a, tb, K,
                p->sel, p->level_cnt, kps_dev, desc_dev, counts_dev, batch, dbg);
    else if (p->steer == 0)
      GH_LAUNCH(ctx, "orb_describe", (describe_kernel<13, false, false>), dim3(8 * gh_div_up(blocks, 8)), dim3(256), p->desc_lds_pad, a, tb, K,
                p->sel, p->level_cnt, kps_dev, desc_dev, counts_dev, batch, dbg);
    else if (p->desc_mfma)  // continuous steering: the 45 x 45 patch's blur on MFMA too (18.7 instead of 38 KB of LDS: 8 workgroups per CU)
      GH_LAUNCH(ctx, "orb_describe", (describe_kernel<19, true, true>), dim3(8 * gh_div_up(blocks, 8)), dim3(256), 0, a, tb, K, p->sel,
                p->level_cnt, kps_dev, desc_dev, counts_dev, batch, dbg);
    else
      GH_LAUNCH(ctx, "orb_describe", (describe_kernel<19, true, false>), dim3(8 * gh_div_up(blocks, 8)), dim3(256), 0, a, tb, K, p->sel,
                p->level_cnt, kps_dev, desc_dev, counts_dev, batch, dbg);
  }
  return GH_OK;
}

#ifdef GH_ORB_PHASES
extern "C" int gh_orb_debug_phases(unsigned int* buf_dev) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_orb_phase_buf), &buf_dev, sizeof(buf_dev)) != hipSuccess;
}
#endif

extern "C" gh_status gh_orb_extract_host(gh_orb_plan* p, const uint8_t* gray, int row_stride, gh_keypoint* kps,
                                         uint8_t* desc, int32_t* count) {
  if (!p) return GH_ERR_ARG;
  gh_ctx* ctx = p->ctx;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, gray && kps && desc && count && row_stride >= p->w);
  const int K = p->prm.n_features;
  // result block: [count, pad to 256 B][K keypoints, padded to 256 B][K descriptors]
  const size_t off_kps = 256, off_desc = off_kps + (((size_t)K * sizeof(gh_keypoint) + 255) & ~(size_t)255);
  const size_t out_bytes = off_desc + (size_t)K * 32;
  // The image goes up as ONE flat copy of h * row_stride bytes (a pitched 2-D copy from pageable memory is issued row by
  // row: measured 2.9 ms for 1241x376 against 0.3 ms for the whole 1080p path); gh_orb_extract_dev takes any stride.
  const size_t img_bytes = (size_t)row_stride * p->h;
  if (!p->stage_img || p->stage_img_bytes < img_bytes + 256) {
    GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (p->stage_img) GH_HIP(ctx, hipFree(p->stage_img));
    p->stage_img = nullptr;
    p->stage_img_bytes = 0;
    GH_TRY(plan_alloc(p, img_bytes + 256, (void**)&p->stage_img));
    p->stage_img_bytes = img_bytes + 256;
  }
  if (!p->stage_out) {
    GH_TRY(plan_alloc(p, out_bytes, (void**)&p->stage_out));
    if (hipHostMalloc((void**)&p->stage_host, out_bytes, hipHostMallocDefault) != hipSuccess) {
      p->stage_host = nullptr;
      return gh_set_error(ctx, GH_ERR_NOMEM, "hipHostMalloc(%zu) for the result staging block failed", out_bytes);
    }
  }
  // the caller's last row may end at its last pixel (ROI view): never read the padding behind it
  GH_HIP(ctx, hipMemcpyAsync(p->stage_img, gray, (size_t)row_stride * (p->h - 1) + p->w, hipMemcpyHostToDevice, ctx->stream));
  GH_TRY(gh_orb_extract_dev(p, p->stage_img, 1, img_bytes, row_stride, reinterpret_cast<gh_keypoint*>(p->stage_out + off_kps),
                            p->stage_out + off_desc, reinterpret_cast<int32_t*>(p->stage_out)));
  GH_HIP(ctx, hipMemcpyAsync(p->stage_host, p->stage_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(count, p->stage_host, sizeof(int32_t));
  memcpy(kps, p->stage_host + off_kps, (size_t)K * sizeof(gh_keypoint));
  memcpy(desc, p->stage_host + off_desc, (size_t)K * 32);
  return GH_OK;
}

extern "C" gh_status gh_bgr_to_gray_dev(gh_ctx* ctx, const uint8_t* bgr_dev, int width, int height, int channels,
                                        int src_row_stride, uint8_t* gray_dev, int dst_row_stride) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, bgr_dev && gray_dev && width > 0 && height > 0 && (channels == 3 || channels == 4));
  GH_CHECK_ARG(ctx, src_row_stride >= width * channels && dst_row_stride >= width);
  GH_LAUNCH(ctx, "bgr_to_gray", bgr_to_gray_kernel, dim3(gh_div_up(width, 256), height), dim3(256), 0, bgr_dev, width,
            height, channels, src_row_stride, (size_t)0, gray_dev, dst_row_stride, (size_t)0);
  return GH_OK;
}

extern "C" gh_status gh_bgr_to_gray_batch_dev(gh_ctx* ctx, const uint8_t* bgr_dev, int width, int height, int channels,
                                              int src_row_stride, size_t src_frame_stride, int n_frames, uint8_t* gray_dev,
                                              int dst_row_stride, size_t dst_frame_stride) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, bgr_dev && gray_dev && width > 0 && height > 0 && height <= 65535 && (channels == 3 || channels == 4));
  GH_CHECK_ARG(ctx, src_row_stride >= width * channels && dst_row_stride >= width && n_frames >= 1 && n_frames <= 65535);
  GH_LAUNCH(ctx, "bgr_to_gray", bgr_to_gray_kernel, dim3(gh_div_up(width, 256), height, n_frames), dim3(256), 0, bgr_dev,
            width, height, channels, src_row_stride, src_frame_stride, gray_dev, dst_row_stride, dst_frame_stride);
  return GH_OK;
}

extern "C" gh_status gh_orb_plan_debug_counters(gh_orb_plan* p, int enable, uint32_t* out16) {
  if (!p) return GH_ERR_ARG;
  gh_ctx* ctx = p->ctx;
  GH_ENTER(ctx);
  if (!p->dbg) {
    GH_TRY(plan_alloc(p, kDbgCount * sizeof(uint32_t), (void**)&p->dbg));
    GH_HIP(ctx, hipMemsetAsync(p->dbg, 0, kDbgCount * sizeof(uint32_t), ctx->stream));
  }
  if (out16) {  // read and clear
    GH_HIP(ctx, hipMemcpyAsync(out16, p->dbg, kDbgCount * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GH_HIP(ctx, hipMemsetAsync(p->dbg, 0, kDbgCount * sizeof(uint32_t), ctx->stream));
  }
  p->dbg_on = enable != 0;
  return GH_OK;
}

extern "C" gh_status gh_orb_debug_level(gh_orb_plan* p, int slot, int level, uint8_t* out_host) {
  if (!p) return GH_ERR_ARG;
  gh_ctx* ctx = p->ctx;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, slot >= 0 && slot < p->max_batch && level >= 1 && level < p->L && out_host);
  GH_HIP(ctx, hipMemcpy2DAsync(out_host, p->lw[level], p->pyr + (size_t)slot * p->slab + p->lvl_off[level],
                               p->pitch[level], p->lw[level], p->lh[level], hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GH_OK;
}
