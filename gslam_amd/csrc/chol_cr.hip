// Banded SPD solve by block cyclic reduction (f64, gfx950): the linear solver of bundle adjustment when the reduced
// camera system is a BAND -- cameras along a trajectory that only share points with their neighbours, the normal case of
// visual odometry / windowed SLAM -- instead of the dense factorisation of chol.hip.
//
// Stands in for the SPARSE_SCHUR solve inside the (absent) Ceres plugin behind GSLAM::Optimizer::optimize
// (GSLAM/core/Optimizer.h:229); the result equals the dense Cholesky solve up to rounding (it IS a Cholesky factorisation,
// of the symmetrically permuted matrix), restated independently in tests/test_cr_solver.py.
//
// Why: the dense single-launch factorisation (ba_potrf_flow) is a chain of n sequential pivots -- 47 diagonal blocks of
// 16 us at n = 3000, 85 % of its wave-cycles idle -- whether or not the tiles off the band are zero.  A band of half-width
// h <= m is block TRIDIAGONAL in superblocks of m = 64 T columns (T = 1..3), and eliminating every other superblock is
// independent work: level r (stride s = 2^r) removes the superblocks i = s (mod 2 s) at once,
//     L_i = chol(D_i),   W_u = B(i,u)^T L_i^-T,  W_d = B(d,i) L_i^-T          (u = i - s, d = i + s)
//     D_u -= W_u W_u^T,  D_d -= W_d W_d^T,  B(d,u) -= W_d W_u^T  (fill: lands in the zero part of the dense lower triangle)
// and the survivors are again block tridiagonal with stride 2 s.  ceil(log2 N) levels + one last block: the chain is
// (levels + 1) m pivots -- 960 instead of 3000 at C4 -- and every level is three launches without any in-launch hand-off:
//     ba_cr_factor   one workgroup per eliminated superblock: potf2 + inverse of its T diagonal tiles (chol_potf2.h),
//                    X = P M^T and the trailing tiles on MFMA, everything of the superblock through LDS / L2
//     ba_cr_panels   W = P L_i^-T by 16-row strips (4 waves = the four 16-column blocks of a tile, the strip in LDS, the
//                    tiles of L_i and M straight from L2 in MFMA operand layout); the right-hand side (row n of A, as in
//                    chol.hip) is one more strip: y_i = b_i L_i^-T
//     ba_cr_update   one workgroup per 64 x 64 tile of D_u / D_d / B(d,u): rank-m update from the W panels (both sources of
//                    a diagonal block in one task: fixed summation order, no atomics); b_j -= y_i W^T for the rhs
// then backwards, one launch per level:  x_i = (y_i - x_u W_u - x_d W_d) L_i^-1   (ba_cr_back).
// Storage: A is the dense column-major lower triangle chol.hip uses (n x n, lda > n, rhs in row n); L_i overwrites D_i, the
// W panels go to a workspace (2 N m^2 doubles), the inverted diagonal tiles to `dinv` (N T tiles).
#include "common.h"
#include "cr_map.h"

#include <type_traits>

// chol.hip: the dense factorisation / back-substitution (the arrowhead solver hands them superblock 0 + the border)
gh_status gh_potrf_dev_impl(gh_ctx* ctx, double* A, int n, int lda, int* info_dev, int extra_rows, double* dinv,
                            double* xwork, unsigned* flow_state, bool store_diag, bool state_ready);
gh_status gh_potrs_bwd_dev_impl(gh_ctx* ctx, const double* L, int n, int lda, double* b, double* work,
                                const double* dinv, const double* yv, long long ystride, double* xh, int* info_dev,
                                bool xh_ready);
size_t gh_potrf_flow_words(const gh_ctx* ctx, int n, int extra_rows);
size_t gh_potrf_flow_flag_words(int n, int extra_rows);
std::mutex& gh_potrf_flow_mutex(int device);
int gh_cr_top(int nbr);

namespace {

constexpr int NBI = 64;
typedef double double4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#include "chol_potf2.h"

// tools/cr_stamp_probe.hip defines GH_CR_PROBE: wall-clock stamps (100 MHz) of workgroup 0 of the level-0 launches
#ifdef GH_CR_PROBE
__device__ long long g_cr_stamp[64];
#define CR_STAMP(slot)                                                                              \
  do {                                                                                              \
    if (a.s == 1 && blockIdx.x == 0 && threadIdx.x == 0) g_cr_stamp[(slot)] = (long long)wall_clock64(); \
  } while (0)
#else
#define CR_STAMP(slot) \
  do {                 \
  } while (0)
#endif

constexpr int kCrXP = 80;  // LDS pitch of the operand tiles of the MFMA phases (see cr_factor_kernel)

struct CrArgs {
  double* A;     // n x n lower triangle + rhs row n, column-major
  int lda, n;
  int N;         // superblocks of m = 64 T columns
  int s;         // stride of this level: eliminated i = first + e * 2 s, neighbours i -+ s
  int first, count;
  double* dinv;  // N * T tiles of 4096 doubles: M = L^-1 of every diagonal tile (column-major, pitch 64, upper part zero)
  double* W;     // [N][2][m * m]: W(i, side)[rho][c] at c * m + rho   (side 0 = u, 1 = d)
  double* G;     // [N][2][m * m]: G(i, side) = W(i, side) L_i^-1, same layout: the backward pass is x_i = yh_i - x_u G_u - x_d G_d
  double* W3;    // [N][m * m]: L_i^-T (the panel of P = identity), same layout
  double* yh;    // [N * m]: yh_i = y_i L_i^-1
  double* x;     // n
  int* info;
  // Arrowhead systems (band + dense border, cr_arrow_solve_t below): `nbr` border rows a.n .. a.n + nbr - 1 lie under the band
  // and the right-hand side is row rr = a.n + nbr (band-only systems: nbr = 0, rr = n -- the layout of rounds 1-4)
  int rr = 0, nbr = 0;
  // DENSE TOP (cr_solve_t): the reduction stops when at most gh_cr_top(nbr) superblocks survive -- 0, keep, 2 keep, ... -- and the
  // block tridiagonal system they form (+ the border) goes to chol.hip's dense path; keep >= N: only superblock 0 survives
  int keep = 1 << 30;
  // STRUCTURE OF THE BORDER (gh_cr_border_symbolic; null = treat E as dense): nzY[i * nbs + strip] != 0 iff the 16-row strip of the
  // border can be non-zero in the columns of superblock i at the moment i is eliminated; nzT[i * ntr + t] the same per 64-row
  // tile of the corner (the tile that holds the right-hand-side row is always set)
  const uint8_t* nzY = nullptr;
  const uint8_t* nzT = nullptr;
  int nbs = 0, ntr = 0;
  // storage of A (cr_map.h): dense (map.m == 0, the C-ABI test entries) or compact (gh_ba_solve)
  CrMap map;
  // base pointers with the shift of the block they address folded in: element (r, c) at ptr[c * lda + r], r / c GLOBAL
  __device__ __forceinline__ double* blk(int I, int J) const { return A + map.shift(I, J); }
  __device__ __forceinline__ double* brd() const { return A + map.bshift(); }
};

// compile-time loop: f(std::integral_constant<int, K>) for K = 0 .. N - 1 (a runtime loop around the potf2 code is not
// unrolled by the compiler, and the accumulator arrays it indexes would then live in scratch)
template <int K, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (K < N) {
    f(std::integral_constant<int, K>{});
    static_for<K + 1, N>(f);
  }
}

// Guarded element of A WITHOUT a branch.  A lane-conditional global load -- and equally a select between a loaded value and
// a constant, which the code generator turns back into a branch around the load -- compiles to "skip if exec is empty; load;
// s_waitcnt vmcnt(0)": every load of a prefetch block waits for the one before it (22 us for the 48 operand loads of a panel
// strip).  So: the address is clamped into the matrix, the load is unconditional, and the value is kept or zeroed by an
// integer AND on its bits (never NaN, whatever the clamped address held).
__device__ __forceinline__ double keep_if(double v, bool ok) {
  return __longlong_as_double(__double_as_longlong(v) & (ok ? ~0ll : 0ll));
}
__device__ __forceinline__ double ld_guard(const double* __restrict__ A, size_t lda, int row, int col, int row_lim, int col_lim) {
  const int rc = row < row_lim ? row : row_lim - 1, cc = col < col_lim ? col : col_lim - 1;
  return keep_if(A[(size_t)cc * lda + rc], row < row_lim && col < col_lim);
}

// D[i][j] += sum_t a(i, t) b(t, j) on v_mfma_f64_16x16x4_f64: the lane supplies a(m, 4 ks + q) and b(4 ks + q, m) and holds
// D[q + 4 r][m] in acc[r]  (m = lane & 15, q = lane >> 4), as lds_mma in chol_potf2.h.
__device__ __forceinline__ double4_t mma(double a, double b, double4_t acc) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
}

// Sum over the wave, valid in lane 63: four DPP row shifts + two row broadcasts (three VALU each for a double), as
// wave_incl_scan_i32 in orb.hip.  Every lane of the wave must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum63(double v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31
  return v;
}

// ------------------------------------------------------------------------------------------------ factor
// One workgroup (8 waves) per eliminated superblock.  Every tile of the superblock is read ONCE, at the start, into MFMA
// accumulators (negated: -A + sum X X^T, lane = [row 16 rg + m][col 16 cb + q + 4 r], the 16 rows of a lane group are 128
// contiguous bytes of A); per diagonal tile k the accumulators of column k go to LDS, potf2 + inverse run there
// (potf2_chain_lds: the inversion hides behind the pivots), X = P M^T and the trailing tiles are MFMA work out of LDS, and
// L / M leave with fire-and-forget stores: no global load sits between two diagonal tiles.
// Wave (rg, hf): rows 16 rg .. of a tile, column blocks 2 hf, 2 hf + 1.
template <int T>
__global__ __launch_bounds__(512) void cr_factor_kernel(CrArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cr_lds[];
  Potf2Lds& sh = *reinterpret_cast<Potf2Lds*>(cr_lds);
  constexpr size_t kShBytes = (sizeof(Potf2Lds) + 15) & ~(size_t)15;
  double* const T2 = reinterpret_cast<double*>(cr_lds + kShBytes);  // 16 x 17 doubles of scratch for the inversion
  // tile j = 1 .. T - 1 of the current column: element (row, t) at t * XP + row.  Pitch 80: the four k-rows (q) of an MFMA
  // operand read start 32 banks apart, so a half-wave touches every bank once (pitch 65 is a two-way conflict)
  constexpr int XP = kCrXP;
  auto xs = [&](int j) {
    return reinterpret_cast<double*>(cr_lds + kShBytes + 16 * 17 * sizeof(double)) + (size_t)(j - 1) * (NBI * XP);
  };
  constexpr int m_ = NBI * T;
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, q = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), rg = wv & 3, hf = wv >> 2;
  const int i = a.first + (int)blockIdx.x * 2 * a.s;
  const int i0 = i * m_, n = a.n;
  const size_t lda = (size_t)a.lda;
  double* const A = a.blk(i, i);      // (every tile of this kernel lies in the superblock's own diagonal block)
  double* const Ab = a.brd();         // the right-hand-side row
  CR_STAMP(0);
  // neg[tile (ii, jj)][cbi]: -(A - sum) of the lower tiles, ii >= jj.  Tile (0, 0) is asked for alone and goes to LDS as
  // soon as it is there; the other tiles are asked for behind that and land while the first potf2 runs (one wait for all 48
  // loads of a lane -- 196 KB through one CU -- was 7.5 us in front of the first pivot).
  double4_t neg[T * (T + 1) / 2][2];
  auto load_tile = [&](int ii, int jj) {
#pragma unroll
    for (int cbi = 0; cbi < 2; ++cbi) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = i0 + NBI * ii + 16 * rg + m, col = i0 + NBI * jj + 16 * (2 * hf + cbi) + q + 4 * r;
        double v = ld_guard(A, lda, row, col, n, n);
        if (ii == jj) v = keep_if(v, col <= row) + ((row == col && row >= n) ? 1.0 : 0.0);  // lower part; identity padding past n
        neg[ii * (ii + 1) / 2 + jj][cbi][r] = -v;
      }
    }
  };
  load_tile(0, 0);
  // The LAST block (no neighbours, a.count == 1 at first == 0) also solves its right-hand side here, both ways: two launches
  // of ~10 us less at the end of the chain.  tvec = b (row n of A), yv = b L^-T tile by tile behind each potf2.
  const bool last = a.first == 0;
  double* const tvec = reinterpret_cast<double*>(cr_lds + kShBytes + 16 * 17 * sizeof(double)) + (size_t)(T - 1) * (NBI * XP);
  double* const yv = tvec + m_;
  if (last && tid < m_) tvec[tid] = ld_guard(Ab, lda, a.rr, i0 + tid, a.rr + 1, n);
  for (int e = tid; e < NBI * LP; e += 512) sh.Ms[e] = 0.0;  // the inversion only ever writes the lower blocks
#pragma unroll
  for (int cbi = 0; cbi < 2; ++cbi) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * rg + m, col = 16 * (2 * hf + cbi) + q + 4 * r;
      sh.As[col * LP + row] = col <= row ? -neg[0][cbi][r] : 0.0;
    }
  }
  asm volatile("" ::: "memory");  // (keeps the requests below behind the LDS writes above: they must not be waited for here)
  // column 0 now (wanted right behind the first potf2), the trailing tiles behind that potf2 (wanted after the X tiles):
  // all of them in flight through the potf2 would not fit the 256 registers of a wave of this workgroup
#pragma unroll
  for (int ii = 1; ii < T; ++ii) load_tile(ii, 0);
  static_for<0, T>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const int c0 = i0 + NBI * k;
    const int kb = n - c0 < NBI ? (n - c0 > 0 ? n - c0 : 0) : NBI;
    double* const Minv = a.dinv + (size_t)(i * T + k) * (NBI * NBI);
    lds_barrier();  // the previous step's products are done with the tile buffers
    if (tid == 0) {
      sh.bad = 0;
      sh.pivots_done = 0;
      sh.next_ready = 0;
      sh.progress = 0;
      sh.x10_done = 0;
    }
    if constexpr (k > 0) {  // (tile (0, 0) went to LDS in the prologue)
#pragma unroll
      for (int cbi = 0; cbi < 2; ++cbi) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * rg + m, col = 16 * (2 * hf + cbi) + q + 4 * r;
          sh.As[col * LP + row] = col <= row ? -neg[k * (k + 1) / 2 + k][cbi][r] : 0.0;
        }
      }
    }
    lds_barrier();
    CR_STAMP(1 + 4 * k);
    potf2_chain_lds(sh, T2, tid, [] {}, [] {});
    CR_STAMP(2 + 4 * k);
    if (tid == 0 && sh.bad) atomicMax(a.info, c0 + 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = tid + 512 * e, c = idx >> 6, r = idx & 63;
      if (r < kb && c < kb && c <= r) A[(size_t)(c0 + c) * lda + c0 + r] = sh.As[c * LP + r];
      Minv[idx] = (c <= r) ? sh.Ms[c * LP + r] : 0.0;
    }
    if (last) {  // y_k = t_k M_kk^T: thread = (column c, the t with t / 8 == wave), partial sums through Ts
      const int c = tid & 63;
      double part = 0.0;
#pragma unroll
      for (int t8 = 0; t8 < 8; ++t8) part = __builtin_fma(sh.Ms[(8 * wv + t8) * LP + c], tvec[NBI * k + 8 * wv + t8], part);
      sh.Ts[wv * 64 + c] = part;
      lds_barrier();
      if (tid < 64) {
        double y = 0.0;
#pragma unroll
        for (int g = 0; g < 8; ++g) y += sh.Ts[g * 64 + tid];
        yv[NBI * k + tid] = y;
      }
      lds_barrier();
    }
    CR_STAMP(3 + 4 * k);
    if constexpr (k == 0) {
#pragma unroll
      for (int ii = 1; ii < T; ++ii) {
#pragma unroll
        for (int jj = 1; jj <= ii; ++jj) load_tile(ii, jj);
      }
    }
    if constexpr (k + 1 < T) {
      // the tiles below, out of their accumulators (for k = 0 their loads had the whole potf2 to land)
#pragma unroll
      for (int cbi = 0; cbi < 2; ++cbi) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * rg + m, col = 16 * (2 * hf + cbi) + q + 4 * r;
#pragma unroll
          for (int j = k + 1; j < T; ++j) xs(j)[col * XP + row] = -neg[j * (j + 1) / 2 + k][cbi][r];
        }
      }
      lds_barrier();
      // X(j, k) = P M^T: one read of P feeds both column blocks of the wave, the chains of different tiles interleave
      double4_t xa[T > 1 ? T - 1 : 1][2];
#pragma unroll
      for (int j = k + 1; j < T; ++j) xa[j - 1][0] = xa[j - 1][1] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (ks < 4 * (2 * hf + 2)) {  // M[c][t] = 0 for t > c
          const double m0 = sh.Ms[(4 * ks + q) * LP + 16 * (2 * hf) + m], m1 = sh.Ms[(4 * ks + q) * LP + 16 * (2 * hf + 1) + m];
#pragma unroll
          for (int j = k + 1; j < T; ++j) {
            const double b = xs(j)[(4 * ks + q) * XP + 16 * rg + m];
            if (ks < 4 * (2 * hf + 1)) xa[j - 1][0] = mma(m0, b, xa[j - 1][0]);
            xa[j - 1][1] = mma(m1, b, xa[j - 1][1]);
          }
        }
      }
      lds_barrier();  // every wave is done reading P
#pragma unroll
      for (int j = k + 1; j < T; ++j) {
        const int r0 = i0 + NBI * j;
#pragma unroll
        for (int cbi = 0; cbi < 2; ++cbi) {
          const int cb = 2 * hf + cbi;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int col = 16 * cb + q + 4 * r, row = 16 * rg + m;
            xs(j)[col * XP + row] = xa[j - 1][cbi][r];
            if (r0 + row < n && col < kb) A[(size_t)(c0 + col) * lda + r0 + row] = xa[j - 1][cbi][r];
          }
        }
      }
      lds_barrier();
      if (last) {  // t_j -= y_k L(j, k)^T for the tiles below: thread = (row r, the c with c / 8 == wave)
        const int r = tid & 63;
#pragma unroll
        for (int j = k + 1; j < T; ++j) {
          double part = 0.0;
#pragma unroll
          for (int c8 = 0; c8 < 8; ++c8) part = __builtin_fma(xs(j)[(8 * wv + c8) * XP + r], yv[NBI * k + 8 * wv + c8], part);
          sh.Ts[(j - k - 1) * 512 + wv * 64 + r] = part;
        }
        lds_barrier();
        if (tid < 64) {
#pragma unroll
          for (int j = k + 1; j < T; ++j) {
            double sum = 0.0;
#pragma unroll
            for (int g = 0; g < 8; ++g) sum += sh.Ts[(j - k - 1) * 512 + g * 64 + tid];
            tvec[NBI * j + tid] -= sum;
          }
        }
        lds_barrier();
      }
      CR_STAMP(4 + 4 * k);
      // trailing tiles (ii, jj), k < jj <= ii < T:  -C += X_ii X_jj^T; per k-step one read of every operand block
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        double av[T][2], bv[T];
#pragma unroll
        for (int j = k + 1; j < T; ++j) {
          bv[j] = xs(j)[(4 * ks + q) * XP + 16 * rg + m];
          av[j][0] = xs(j)[(4 * ks + q) * XP + 16 * (2 * hf) + m];
          av[j][1] = xs(j)[(4 * ks + q) * XP + 16 * (2 * hf + 1) + m];
        }
#pragma unroll
        for (int jj = k + 1; jj < T; ++jj) {
#pragma unroll
          for (int ii = jj; ii < T; ++ii) {
#pragma unroll
            for (int cbi = 0; cbi < 2; ++cbi) {
              if (ii == jj && 2 * hf + cbi > rg) continue;  // strictly upper 16 x 16 blocks of a diagonal tile
              neg[ii * (ii + 1) / 2 + jj][cbi] = mma(av[jj][cbi], bv[ii], neg[ii * (ii + 1) / 2 + jj][cbi]);
            }
          }
        }
      }
    }
  });
  if (last) {
    // x = y L^-1, tile by tile from the last: x_k = (y_k - sum_{j > k} x_j L(j, k)) M_kk.  In LDS at this point: M of the last
    // tile (Ms), L(j, j - 1) in xs(j); the other tiles come back from L2 (this CU stored them a moment ago).  A wave per
    // column, lanes along it (wave_sum63); x overwrites yv.
    static_assert((T - 1) * 512 * sizeof(double) <= sizeof(sh.Ts) || T <= 2, "Ts holds the partial sums of T - 1 tiles");
    double* const xv = yv;
    auto reload = [&](double* dst, int pitch, const double* src, size_t ld, bool guard, int r0, int c0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = tid + 512 * e, c = idx >> 6, r = idx & 63;
        dst[c * pitch + r] = guard ? ld_guard(src, ld, r0 + r, c0 + c, n, n) : src[(size_t)c * ld + r];
      }
    };
    for (int k = T - 1; k >= 0; --k) {
      lds_barrier();
      if (k < T - 1) reload(sh.Ms, LP, a.dinv + (size_t)(i * T + k) * (NBI * NBI), NBI, false, 0, 0);
      for (int j = k + 2; j < T; ++j) {  // (T = 3: only L(2, 0)) through As, one tile at a time
        reload(sh.As, LP, A, lda, true, i0 + NBI * j, i0 + NBI * k);
        lds_barrier();
        for (int c = wv; c < NBI; c += 8) {
          const double sum = wave_sum63(xv[NBI * j + lane] * sh.As[c * LP + lane]);
          if (lane == 63) yv[NBI * k + c] -= sum;
        }
        lds_barrier();
      }
      lds_barrier();
      if (k + 1 < T) {
        for (int c = wv; c < NBI; c += 8) {
          const double sum = wave_sum63(xv[NBI * (k + 1) + lane] * xs(k + 1)[c * XP + lane]);
          if (lane == 63) yv[NBI * k + c] -= sum;
        }
        lds_barrier();
      }
      double xk[8];
#pragma unroll
      for (int cj = 0; cj < 8; ++cj) xk[cj] = wave_sum63(yv[NBI * k + lane] * sh.Ms[(wv + 8 * cj) * LP + lane]);  // M[r][c], zero above the diagonal
      lds_barrier();  // every wave has read y_k
#pragma unroll
      for (int cj = 0; cj < 8; ++cj)
        if (lane == 63) xv[NBI * k + wv + 8 * cj] = xk[cj];
    }
    lds_barrier();
    if (tid < m_ && i0 + tid < n) a.x[i0 + tid] = xv[tid];
  }
  CR_STAMP(15);
}

// ------------------------------------------------------------------------------------------------ panels
// One workgroup (4 waves) per 16-row strip of a panel of an eliminated superblock i: W = P L_i^-T, column tile by column
// tile:  W_c = (P_c - sum_{c' < c} W_c' L(c, c')^T) M_cc^T.  Wave w owns the 16-column block w of every tile.  Every tile of
// L_i / M the wave will multiply with is asked for at the START, in MFMA operand layout (the 16 rows of a lane group are
// contiguous), together with the strip itself: one memory round trip, then only LDS and MFMA phases.  The strip lives in LDS
// as [column t][row j] (pitch 17): Pl holds P / P', Xl the finished W.
// task: 0 .. 4T-1 strips of side u (P = B(i, u)^T), 4T .. 8T-1 side d (P = B(d, i)), 8T = the right-hand side,
// 8T+1 .. 12T strips of P = identity: W3 = L_i^-T, what turns the backward pass into plain products (cr_update_kernel).
template <int T>
__global__ __launch_bounds__(256) void cr_panels_kernel(CrArgs a, int inverse, int ntask) {
  constexpr int m_ = NBI * T, PW = 17;
  __shared__ double Pl[m_ * PW], Xl[m_ * PW];
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // every task of an eliminated superblock on ONE XCD (block b runs on XCD b % 8, each with its own L2): the six operand
  // tiles come over the fabric once per XCD instead of once per workgroup (placement only affects speed)
  const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  const int e = xcd + 8 * (slot / ntask), task = slot % ntask;
  if (e >= a.count) return;
  // (a.s == 0: every eliminated superblock of every level, i = 1 .. N - 1 -- only the identity strips are run that way)
  const int i = a.s == 0 ? 1 + e : a.first + e * 2 * a.s, i0 = i * m_, n = a.n;
  if (a.s == 0 && (i & (a.keep - 1)) == 0) return;  // a survivor of the dense top: never eliminated, no L_i
  const size_t lda = (size_t)a.lda;
  const double* const A = a.blk(i, i);  // L_i and M of the eliminated superblock
  // forward launches: tasks 0 .. 4T-1 side u, 4T .. 8T-1 side d, 8T the right-hand side, from 8T + 1 side 4 = a 16-row strip of
  // the BORDER rows a.n .. a.n + nbr - 1 of an arrowhead system (Y_i = E_i L_i^-T in place); inverse launches: 4T strips of
  // the identity (side 3)
  const int side = inverse ? 3 : (task > 8 * T ? 4 : (task == 8 * T ? 2 : (task >= 4 * T ? 1 : 0)));
  const int strip = side == 4 ? task - 8 * T - 1 : (side == 3 ? task : task - 4 * T * (side == 1 ? 1 : (side == 2 ? 2 : 0)));
  const int nb = side == 0 ? i - a.s : i + a.s;
  if (side < 2 && (nb < 0 || nb >= a.N)) return;
  if (side == 4 && a.nzY != nullptr && a.nzY[i * a.nbs + strip] == 0) return;  // E_i is zero in these rows: Y_i = 0 is already there
  const int nb0 = nb * m_ + 16 * strip;  // first row of the strip (side u / d)
  // the strip's source: B(i, u) (rows of i in the columns of u), B(d, i), the right-hand side / a border strip, or nothing (identity)
  const double* const Ap = side == 0 ? a.blk(i, nb) : (side == 1 ? a.blk(nb, i) : ((side == 2 || side == 4) ? a.brd() : a.blk(0, 0)));
  CR_STAMP(16);
  // ---- the strip of P: asked for first (it is needed first), into registers; LDS behind the operand requests below
  constexpr int kStripIt = 16 * m_ / 256;
  double pv[kStripIt];
#pragma unroll
  for (int it = 0; it < kStripIt; ++it) {
    // (one branch-free load per element whatever the side: see ld_guard)
    const int idx = tid + 256 * it;
    const int j = side == 0 ? idx / m_ : (idx & 15), t = side == 0 ? idx - (idx / m_) * m_ : (idx >> 4);
    const int row = side == 0 ? i0 + t : (side == 1 ? nb0 + j : (side == 2 ? a.rr : (side == 4 ? n + 16 * strip + j : 0)));
    const int col = side == 0 ? nb0 + j : (side == 3 ? 0 : i0 + t);
    const double v = ld_guard(Ap, lda, row, col, side == 2 ? a.rr + 1 : (side == 4 ? n + a.nbr : n), n);
    pv[it] = keep_if(v, side < 2 || (side == 2 && j == 0) || side == 4) + ((side == 3 && t == 16 * strip + j) ? 1.0 : 0.0);
  }
  // ---- operands of every product of this wave
  double mreg[T][16], lreg[T * (T - 1) / 2 > 0 ? T * (T - 1) / 2 : 1][16];
#pragma unroll
  for (int c = 0; c < T; ++c) {
    const double* Minv = a.dinv + (size_t)(i * T + c) * (NBI * NBI);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) mreg[c][ks] = Minv[(4 * ks + q) * NBI + 16 * w + m];  // (stored zeros above the diagonal)
#pragma unroll
    for (int cp = 0; cp < c; ++cp) {
      const int lrow = i0 + NBI * c + 16 * w + m;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const int lcol = i0 + NBI * cp + 4 * ks + q;
        lreg[c * (c - 1) / 2 + cp][ks] = -ld_guard(A, lda, lrow, lcol, n, n);
      }
    }
  }
#pragma unroll
  for (int it = 0; it < kStripIt; ++it) {
    const int idx = tid + 256 * it;
    const int j = side == 0 ? idx / m_ : (idx & 15), t = side == 0 ? idx - (idx / m_) * m_ : (idx >> 4);
    Pl[t * PW + j] = pv[it];
  }
  lds_barrier();
  CR_STAMP(17);
#pragma unroll
  for (int c = 0; c < T; ++c) {
    if (c > 0) {
      // P'_c = P_c - sum_{c' < c} W_c' L(c, c')^T : lane holds [row m][col 64 c + 16 w + q + 4 r]; two accumulation chains
      double4_t acc0, acc1 = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < 4; ++r) acc0[r] = Pl[(NBI * c + 16 * w + q + 4 * r) * PW + m];
#pragma unroll
      for (int cp = 0; cp < c; ++cp) {
#pragma unroll
        for (int ks = 0; ks < 16; ks += 2) {
          acc0 = mma(lreg[c * (c - 1) / 2 + cp][ks], Xl[(NBI * cp + 4 * ks + q) * PW + m], acc0);
          acc1 = mma(lreg[c * (c - 1) / 2 + cp][ks + 1], Xl[(NBI * cp + 4 * (ks + 1) + q) * PW + m], acc1);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) Pl[(NBI * c + 16 * w + q + 4 * r) * PW + m] = acc0[r] + acc1[r];
      lds_barrier();
    }
    // W_c = P'_c M_cc^T
    double4_t x0 = (double4_t){0.0, 0.0, 0.0, 0.0}, x1 = x0;
#pragma unroll
    for (int ks = 0; ks < 16; ks += 2) {
      if (ks < 4 * (w + 1)) {
        x0 = mma(mreg[c][ks], Pl[(NBI * c + 4 * ks + q) * PW + m], x0);
        x1 = mma(mreg[c][ks + 1], Pl[(NBI * c + 4 * (ks + 1) + q) * PW + m], x1);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Xl[(NBI * c + 16 * w + q + 4 * r) * PW + m] = x0[r] + x1[r];
    lds_barrier();
  }
  CR_STAMP(18);
  // ---- out
  if (side == 4) {
#pragma unroll
    for (int it = 0; it < 16 * m_ / 256; ++it) {
      const int idx = tid + 256 * it, j = idx & 15, t = idx >> 4;
      if (i0 + t < n && 16 * strip + j < a.nbr) a.brd()[(size_t)(i0 + t) * lda + n + 16 * strip + j] = Xl[t * PW + j];
    }
  } else if (side != 2) {
    double* Wo = (side == 3 ? a.W3 + (size_t)i * ((size_t)m_ * m_) : a.W + ((size_t)i * 2 + side) * ((size_t)m_ * m_)) + 16 * strip;
#pragma unroll
    for (int it = 0; it < 16 * m_ / 256; ++it) {
      const int idx = tid + 256 * it, j = idx & 15, t = idx >> 4;
      Wo[(size_t)t * m_ + j] = Xl[t * PW + j];
    }
  } else {
    for (int t = tid; t < m_; t += 256)
      if (i0 + t < n) a.brd()[(size_t)(i0 + t) * lda + a.rr] = Xl[t * PW];
  }
  CR_STAMP(19);
}

// ------------------------------------------------------------------------------------------------ update
// Products of the panels the previous launch wrote; one 16 x 16 block of a 64 x 64 output tile per WAVE, operands read
// straight from L2 in MFMA layout (no LDS, no barrier): every task of a group runs on ONE XCD (block b -> XCD b % 8), so a
// panel crosses the fabric once per XCD and the 16-fold re-reads of its rows hit that XCD's L2.
// mode 0, group = survivor j = 2 s q:  tiles 0 .. T(T+1)/2 - 1  the lower tiles of D_j (sources: W_d of j - s, W_u of j + s),
//   then T T tiles of the fill block B(j + 2 s, j) (source j + s: W_d W_u^T), then the right-hand side of j (m / 16 workgroups).
// mode 1, group = eliminated i = 1 + group of ANY level (off the critical path: cr_solve_t runs it once, on a side stream,
//   under the factorisation of the last block): 2 T T tiles of
//   G(i, side) = W(i, side) L_i^-1 = W W3^T (W3 = L_i^-T is upper triangular: only the K-tiles >= the column tile count), then
//   yh_i = y_i L_i^-1.
// A workgroup = one quadrant (32 x 32) of a tile, or the vector task of the group.
template <int T>
__device__ __forceinline__ void cr_update_body(const CrArgs& a, int mode, int ngroups, int bid) {
  constexpr int m_ = NBI * T, ND = T * (T + 1) / 2, NV = m_ / 16, NTS = 4 * (ND + T * T) + NV, NTE = 4 * (2 * T * T) + NV;
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = a.n;
  const size_t lda = (size_t)a.lda;
  const size_t mm = (size_t)m_ * m_;
  const bool elim_task = mode != 0;
  const int ntask = elim_task ? NTE : NTS;
  // groups -> XCDs: eight groups share the eight XCDs one each; FEWER groups (the upper levels) spread each group's tasks
  // over 8 / G XCDs instead of crowding 60-70 workgroups onto the 32 CUs of one (the panels then cross the fabric 8 / G
  // times: nothing at those sizes)
  const int G = ngroups >= 8 ? 8 : (ngroups > 4 ? 8 : (ngroups > 2 ? 4 : (ngroups > 1 ? 2 : 1))), kx = 8 / G;
  const int per = (ntask + kx - 1) / kx;  // slots per XCD and group
  const int xcd = bid & 7, slot = bid >> 3;
  const int grp = xcd / kx + G * (slot / per), task = (slot % per) * kx + xcd % kx;
  if (grp >= ngroups || task >= ntask) return;
  // ---- the vector tasks: out[c] (+)= -+ sum_t y[t] W[c][t], 16 columns per workgroup (a whole panel through one CU takes
  // ~25 us); thread = (column tid & 15, the t with t % 16 == tid >> 4): every wave instruction reads four 128-byte rows
  if (task >= ntask - NV) {
    __shared__ double red[16][17];
    const int c = 16 * (task - (ntask - NV)) + (tid & 15), tg = tid >> 4;
    const int j = grp * 2 * a.s;             // survivor
    const int i = elim_task ? 1 + grp : 0;   // eliminated (mode 1 runs over the superblocks of every level)
    if (elim_task && (i & (a.keep - 1)) == 0) return;  // (a survivor of the dense top; uniform, before the barrier)
    double sum = 0.0;
#pragma unroll
    for (int sd = 0; sd < 2; ++sd) {
      int src;
      const double* Wp;
      if (elim_task) {
        if (sd == 1) continue;
        src = i;
        Wp = a.W3 + (size_t)i * mm;
      } else {
        src = sd == 0 ? j - a.s : j + a.s;
        if (src < 0 || src >= a.N) continue;
        Wp = a.W + ((size_t)src * 2 + (sd == 0 ? 1 : 0)) * mm;  // j is the d-neighbour of j - s, the u-neighbour of j + s
      }
      const int s0 = src * m_;
      double yv[m_ / 16], wv[m_ / 16];
#pragma unroll
      for (int k = 0; k < m_ / 16; ++k) {
        const int t = tg + 16 * k;
        yv[k] = ld_guard(a.brd(), lda, a.rr, s0 + t, a.rr + 1, n);
        wv[k] = Wp[(size_t)t * m_ + c];
      }
#pragma unroll
      for (int k = 0; k < m_ / 16; ++k) sum = __builtin_fma(yv[k], wv[k], sum);
    }
    red[tg][tid & 15] = sum;
    __syncthreads();
    if (tid < 16) {
      double tot = 0.0;
#pragma unroll
      for (int g = 0; g < 16; ++g) tot += red[g][tid];
      if (elim_task) a.yh[(size_t)i * m_ + c] = tot;
      else if (j * m_ + c < n) a.brd()[(size_t)(j * m_ + c) * lda + a.rr] -= tot;
    }
    return;
  }
  // ---- tile tasks
  const int tile = task >> 2, quad = task & 3;
  const int rb = 2 * (quad & 1) + (w & 1), cb = 2 * (quad >> 1) + (w >> 1);  // this wave's 16 x 16 block of the tile
  int ta, tb, tt_lo = 0, nsrc = 0, row0, col0;
  bool diag = false;
  const double *R0 = nullptr, *R1 = nullptr, *C0 = nullptr, *C1 = nullptr;  // panel of the row / column operand per source
  double* out;      // element (row, col) of the output tile at out[col * ldo + row]
  size_t ldo;
  if (elim_task) {
    const int i = 1 + grp, si = i & -i;  // eliminated at the level of stride si
    if ((i & (a.keep - 1)) == 0) return;         // (a survivor of the dense top)
    const int side = tile / (T * T), f = tile - side * (T * T);
    const int nb = side == 0 ? i - si : i + si;
    if (nb < 0 || nb >= a.N) return;
    ta = f / T;
    tb = f - ta * T;
    tt_lo = tb;
    nsrc = 1;
    R0 = a.W + ((size_t)i * 2 + side) * mm;
    C0 = a.W3 + (size_t)i * mm;
    out = a.G + ((size_t)i * 2 + side) * mm + (size_t)(NBI * tb) * m_ + NBI * ta;
    ldo = m_;
    row0 = col0 = 0;  // (no bounds: the panels are padded with zeros)
  } else {
    const int j = grp * 2 * a.s;
    const int src_lo = j - a.s, src_hi = j + a.s;  // eliminated neighbours of j (each may be absent)
    const bool has_lo = src_lo >= 0, has_hi = src_hi < a.N;
    int row_blk;
    if (tile < ND) {
      ta = tile < 1 ? 0 : (tile < 3 ? 1 : 2);
      tb = tile - ta * (ta + 1) / 2;
      row_blk = j;
      if (has_lo) {
        R0 = C0 = a.W + ((size_t)src_lo * 2 + 1) * mm;
        nsrc = 1;
      }
      if (has_hi) {
        (nsrc ? R1 : R0) = a.W + ((size_t)src_hi * 2 + 0) * mm;
        (nsrc ? C1 : C0) = a.W + ((size_t)src_hi * 2 + 0) * mm;
        ++nsrc;
      }
      diag = ta == tb;
    } else {
      const int f = tile - ND;
      ta = f / T;
      tb = f - ta * T;
      row_blk = j + 2 * a.s;
      if (row_blk >= a.N) return;
      nsrc = 1;
      R0 = a.W + ((size_t)src_hi * 2 + 1) * mm;  // W_d of j + s: rows of j + 2 s
      C0 = a.W + ((size_t)src_hi * 2 + 0) * mm;  // W_u of j + s: rows of j
    }
    if (nsrc == 0) return;
    row0 = row_blk * m_ + NBI * ta;
    col0 = j * m_ + NBI * tb;
    if (row0 >= n || col0 >= n) return;
    out = a.blk(row_blk, j) + (size_t)col0 * lda + row0;  // D_j, or the fill block B(j + 2 s, j)
    ldo = lda;
  }
  if (diag && cb > rb) return;  // strictly upper block of a diagonal tile
  const int rlim = elim_task ? NBI : n - row0, clim = elim_task ? NBI : n - col0;  // valid rows / columns of the output tile
  CR_STAMP(24);
  double4_t acc0, acc1 = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * rb + m, col = 16 * cb + q + 4 * r;
    double v = 0.0;
    if (!elim_task) v = ld_guard(out, ldo, row, col, rlim, clim);  // (uniform branch)
    acc0[r] = -keep_if(v, !(diag && col > row));
  }
  CR_STAMP(25);
  // the phases (source, K-tile) of this task are the contiguous range [tt_lo, nsrc T); the operands of phase p + 1 are
  // asked for before the MFMAs of phase p (two register sets, ping-pong)
  const int p_hi = nsrc * T;
  auto fetch = [&](int ph, double (&av)[16], double (&bv)[16]) {
    const int sd = ph >= T ? 1 : 0, tt = ph - sd * T;
    const double* Rp = (sd == 0 ? R0 : R1) + (size_t)(NBI * tt + q) * m_ + NBI * ta + 16 * rb + m;
    const double* Cp = (sd == 0 ? C0 : C1) + (size_t)(NBI * tt + q) * m_ + NBI * tb + 16 * cb + m;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      av[ks] = Cp[(size_t)(4 * ks) * m_];
      bv[ks] = Rp[(size_t)(4 * ks) * m_];
    }
  };
  auto product = [&](const double (&av)[16], const double (&bv)[16]) {
#pragma unroll
    for (int ks = 0; ks < 16; ks += 2) {
      acc0 = mma(av[ks], bv[ks], acc0);
      acc1 = mma(av[ks + 1], bv[ks + 1], acc1);
    }
  };
  {
    double a0[16], b0[16], a1[16], b1[16];
    int ph = tt_lo;
    fetch(ph, a0, b0);
    while (true) {
      if (ph + 1 < p_hi) fetch(ph + 1, a1, b1);
      product(a0, b0);
      if (++ph >= p_hi) break;
      if (ph + 1 < p_hi) fetch(ph + 1, a0, b0);
      product(a1, b1);
      if (++ph >= p_hi) break;
    }
  }
  CR_STAMP(26);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * rb + m, col = 16 * cb + q + 4 * r;
    const double v = acc0[r] + acc1[r];
    if (row < rlim && col < clim && (!diag || col <= row)) out[(size_t)col * ldo + row] = elim_task ? v : -v;
  }
  CR_STAMP(27);
}

template <int T>
__global__ __launch_bounds__(256) void cr_update_kernel(CrArgs a, int mode, int ngroups) {
  cr_update_body<T>(a, mode, ngroups, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------------ backward
// Levels: x_i = yh_i - x_u G_u - x_d G_d.  16 columns of a superblock per workgroup (49 KB of G: a workgroup that streams
// a whole superblock's panels -- 590 KB -- through one CU takes ~25 us), a wave per 4 columns, lanes along the column.
template <int T>
__global__ __launch_bounds__(256) void cr_back_kernel(CrArgs a) {
  constexpr int m_ = NBI * T, NG = m_ / 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  const int e = xcd + 8 * (slot / NG), g = slot % NG;
  if (e >= a.count) return;
  const int i = a.first + e * 2 * a.s, i0 = i * m_, n = a.n;
  const size_t mm = (size_t)m_ * m_;
  const int u = i - a.s, d = i + a.s;
  const bool has_u = u >= 0, has_d = d < a.N;
  CR_STAMP(32);
  double xu[T], xd[T];
#pragma unroll
  for (int rr = 0; rr < T; ++rr) {
    const int rho = lane + 64 * rr;
    const int iu = has_u ? u * m_ + rho : 0, id = has_d ? d * m_ + rho : 0;
    xu[rr] = keep_if(a.x[iu < n ? iu : n - 1], has_u && iu < n);
    xd[rr] = keep_if(a.x[id < n ? id : n - 1], has_d && id < n);
  }
  const double* Gu = a.G + ((size_t)i * 2 + 0) * mm;
  const double* Gd = a.G + ((size_t)i * 2 + 1) * mm;
  double part[4];
#pragma unroll
  for (int cj = 0; cj < 4; ++cj) {
    const int c = 16 * g + 4 * wv + cj;
    double sum = 0.0;
#pragma unroll
    for (int rr = 0; rr < T; ++rr) {
      const int rho = lane + 64 * rr;
      // (a missing neighbour's panel was never written: its bits are dropped, not multiplied by zero)
      sum = __builtin_fma(xu[rr], keep_if(Gu[(size_t)c * m_ + rho], has_u), sum);
      sum = __builtin_fma(xd[rr], keep_if(Gd[(size_t)c * m_ + rho], has_d), sum);
    }
    part[cj] = sum;
  }
  CR_STAMP(33);
#pragma unroll
  for (int cj = 0; cj < 4; ++cj) {
    const int c = 16 * g + 4 * wv + cj;
    const double sum = wave_sum63(part[cj]);
    if (lane == 63 && i0 + c < n) a.x[i0 + c] = a.yh[(size_t)i * m_ + c] - sum;
  }
  CR_STAMP(36);
}

// The last block (no neighbours): x = y L^-1 by one workgroup, a wave per column, lanes along it; the columns of L / M the
// triangular solve needs are asked for before anything else.
__device__ __forceinline__ double wave_sum(double v) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(wave_sum63(v)), 63),
                          __builtin_amdgcn_readlane(__double2loint(wave_sum63(v)), 63));
}
template <int T>
__global__ __launch_bounds__(1024) void cr_back_last_kernel(CrArgs a) {
  constexpr int m_ = NBI * T;
  __shared__ double xu[m_], xd[m_], tv[m_], zv[m_];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = a.first + (int)blockIdx.x * 2 * a.s, i0 = i * m_, n = a.n;
  const size_t lda = (size_t)a.lda;
  const size_t mm = (size_t)m_ * m_;
  const int u = i - a.s, d = i + a.s;
  const bool has_u = u >= 0, has_d = d < a.N;
  // columns wv + 16 cj of every tile: M_kk[lane][c] and L(kp, k)[lane][c]
  double mreg[T][4], lreg[T * (T - 1) / 2 > 0 ? T * (T - 1) / 2 : 1][4];
#pragma unroll
  for (int k = 0; k < T; ++k) {
    const double* Minv = a.dinv + (size_t)(i * T + k) * (NBI * NBI);
#pragma unroll
    for (int cj = 0; cj < 4; ++cj) {
      const int c = wv + 16 * cj;
      mreg[k][cj] = Minv[c * NBI + lane];  // zero above the diagonal
#pragma unroll
      for (int kp = k + 1; kp < T; ++kp) {
        const int row = i0 + NBI * kp + lane, col = i0 + NBI * k + c;
        lreg[kp * (kp - 1) / 2 + k][cj] = ld_guard(a.blk(i, i), lda, row, col, n, n);
      }
    }
  }
  if (tid < m_) {
    xu[tid] = (has_u && u * m_ + tid < n) ? a.x[u * m_ + tid] : 0.0;
    xd[tid] = (has_d && d * m_ + tid < n) ? a.x[d * m_ + tid] : 0.0;
    tv[tid] = i0 + tid < n ? a.brd()[(size_t)(i0 + tid) * lda + a.rr] : 0.0;
  }
  __syncthreads();
  if (has_u || has_d) {
    const double* Wu = a.W + ((size_t)i * 2 + 0) * mm;
    const double* Wd = a.W + ((size_t)i * 2 + 1) * mm;
    double part[m_ / 16];
#pragma unroll
    for (int ci = 0; ci < m_ / 16; ++ci) {
      const int c = wv + 16 * ci;
      double sum = 0.0;
#pragma unroll
      for (int rr = 0; rr < T; ++rr) {
        const int rho = lane + 64 * rr;
        if (has_u) sum = __builtin_fma(xu[rho], Wu[(size_t)c * m_ + rho], sum);
        if (has_d) sum = __builtin_fma(xd[rho], Wd[(size_t)c * m_ + rho], sum);
      }
      part[ci] = sum;
    }
#pragma unroll
    for (int ci = 0; ci < m_ / 16; ++ci) {
      const double sum = wave_sum(part[ci]);
      if (lane == 0) tv[wv + 16 * ci] -= sum;
    }
    __syncthreads();
  }
  // z L_i = t, tile by tile from the last:  z_k = (t_k - sum_{k' > k} z_k' L(k', k)) M_kk
#pragma unroll
  for (int k = T - 1; k >= 0; --k) {
    if (k < T - 1) {
#pragma unroll
      for (int cj = 0; cj < 4; ++cj) {
        double sum = 0.0;
#pragma unroll
        for (int kp = k + 1; kp < T; ++kp) sum = __builtin_fma(zv[NBI * kp + lane], lreg[kp * (kp - 1) / 2 + k][cj], sum);
        sum = wave_sum(sum);
        if (lane == 0) tv[NBI * k + wv + 16 * cj] -= sum;
      }
      __syncthreads();
    }
#pragma unroll
    for (int cj = 0; cj < 4; ++cj) {
      const double sum = wave_sum(tv[NBI * k + lane] * mreg[k][cj]);
      if (lane == 0) zv[NBI * k + wv + 16 * cj] = sum;
    }
    __syncthreads();
  }
  if (tid < m_ && i0 + tid < n) a.x[i0 + tid] = zv[tid];
}

// ------------------------------------------------------------------------------------------------ arrowhead: the border
// A reduced camera system whose cameras follow a trajectory EXCEPT for a few long-range points (loop closures) is a band plus a
// dense border once the cameras those points tie to far-away ones are numbered last (ba.hip: arrow ordering):
//     [ B   E^T ] [x_b]   [g_b]        B: band, n x n (a.n)         rows 0 .. n - 1
//     [ E   C   ] [x_c] = [g_c]        E: nbr x n, C: nbr x nbr     rows n .. n + nbr - 1;  right-hand side: row rr = n + nbr
// Block cyclic reduction runs on B exactly as above, with E and the right-hand side riding along as EXTRA ROWS of every
// eliminated superblock i:  Y_i = E_i L_i^-T (cr_panels_kernel, side 4), E_u -= Y_i W_u^T, E_d -= Y_i W_d^T
// (cr_border_update_kernel) -- what the band solver does to its one right-hand-side row, for nbr rows.  When only superblock 0
// is left, C has to lose sum_i Y_i Y_i^T over every eliminated i (cr_border_syrk_kernel: the rank-(n - m) update of the corner,
// right-hand-side row included, in K chunks whose partial tiles are summed in a fixed order), and what remains is the DENSE
// system of superblock 0 and the border, (m + nbr)^2, solved by chol.hip's dense path.  Backwards the border only adds
// - x_c Y_i to the right-hand side of superblock i:  yh_i -= (x_c Y_i) L_i^-1, then cr_back_kernel as before.
// It is a Cholesky factorisation of the symmetrically permuted matrix: same result as the dense solve to rounding
// (tests/test_cr_solver.py: restated in numpy; gh_arrow_solve_dev against numpy and the dense path).
// K chunks of the corner update: columns per chunk (a multiple of 16) and their number -- at most 16 superblocks per chunk, and
// enough chunks that tiles x chunks fill the chip when the corner has few tiles (C4 + 20 loop closures: 21 tiles)
// The partial tiles take chunks x tiles x 32 KB: bounded to kBorderPartTiles tiles (512 MB) -- a wide border (1024 cameras: 4753
// tiles) gets few, long chunks instead of 2.8 GB of partial sums.
constexpr int kBorderPartTiles = 16384;
struct BorderChunks { int kc, n; };
inline BorderChunks border_chunks(int n_band, int m, int nbr) {
  const int K = n_band > m ? n_band : 0;  // (the chunks run over every band column; the survivors' columns are skipped in the kernel)
  if (K == 0 || nbr == 0) return BorderChunks{16, 0};
  const int ntr = (nbr + 1 + 63) / 64, ntiles = ntr * (ntr + 1) / 2;
  int want = (K + 16 * m - 1) / (16 * m);
  const int fill = (1024 + ntiles - 1) / ntiles;
  if (fill > want) want = fill < 64 ? fill : 64;
  const int cap = kBorderPartTiles / ntiles > 1 ? kBorderPartTiles / ntiles : 1;
  if (want > cap) want = cap;
  if (want > (K + 63) / 64) want = (K + 63) / 64;
  const int kc = (((K + want - 1) / want) + 15) & ~15;
  return BorderChunks{kc, (K + kc - 1) / kc};
}

// level s, survivor j = grp 2 s:  E_j -= Y_{j-s} W_d(j-s)^T + Y_{j+s} W_u(j+s)^T.  Workgroup = 64 border rows x 16 columns of j,
// wave = one 16 x 16 block, operands straight from L2 in MFMA layout with ALL k-steps of a source in flight at once (the launch
// is a chain of L2 round trips: two per wave this way; 32 x 32 blocks with 16 k-steps in flight took 35 us per level at C4).
// D'[out column][out row]: the lane holds out column q + 4 r, out row m -- 16 consecutive rows of A per register, a 128-byte run.
template <int T>
__device__ __forceinline__ void cr_border_update_body(const CrArgs& a, int nrt, int bid) {
  constexpr int m_ = NBI * T, KS = m_ / 4;
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per = nrt * T * 4;
  const int grp = bid / per, rem = bid - grp * per, rt = rem / (T * 4), cq = rem - rt * (T * 4);
  const int j = grp * 2 * a.s, n = a.n;
  if (j >= a.N) return;
  const size_t lda = (size_t)a.lda, mm = (size_t)m_ * m_;
  const double* const A = a.brd();     // (border rows only)
  const int row_b = 64 * rt + 16 * w;  // border row of this wave's block
  const int col_b = 16 * cq;           // column within superblock j
  if (row_b >= a.nbr) return;
  const int r = row_b + m;
  const size_t roff = (size_t)n + (size_t)(r < a.nbr ? r : a.nbr - 1);
  double4_t acc0 = (double4_t){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
  bool any = false;
#pragma unroll
  for (int sd = 0; sd < 2; ++sd) {
    const int src = sd == 0 ? j - a.s : j + a.s;
    if (src < 0 || src >= a.N) continue;
    if (a.nzY != nullptr && a.nzY[src * a.nbs + (row_b >> 4)] == 0) continue;  // (wave-uniform) Y_src is zero in this wave's rows
    any = true;
    const double* Wp = a.W + ((size_t)src * 2 + (sd == 0 ? 1 : 0)) * mm + col_b + m;  // j is the d-neighbour of j - s, the u-neighbour of j + s
    const int s0 = src * m_;
    double av[KS], bv[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = 4 * ks + q;
      av[ks] = Wp[(size_t)k * m_];  // (panels are zero-padded)
      const int kc = s0 + k < n ? s0 + k : n - 1;
      bv[ks] = keep_if(A[(size_t)kc * lda + roff], r < a.nbr && s0 + k < n);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2) {
      acc0 = mma(av[ks], bv[ks], acc0);
      acc1 = mma(av[ks + 1], bv[ks + 1], acc1);
    }
  }
  if (!any) return;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int col = j * m_ + col_b + q + 4 * rr;
    if (col < n && r < a.nbr) a.brd()[(size_t)col * lda + n + r] -= acc0[rr] + acc1[rr];
  }
}

template <int T>
__global__ __launch_bounds__(256) void cr_border_update_kernel(CrArgs a, int nrt) {
  cr_border_update_body<T>(a, nrt, (int)blockIdx.x);
}
// Both updates of a level in ONE launch: the survivors' tiles (cr_update_body, the first n_update workgroups) and the border rows
// of the survivors (cr_border_update_body).  They write disjoint parts of A from the same panels; as two launches the second cost
// a dependent-launch latency per level (~10 us: a third of what a 60-row border adds to a C4 iteration).
template <int T>
__global__ __launch_bounds__(256) void cr_update_both_kernel(CrArgs a, int ngroups, int n_update, int nrt) {
  if ((int)blockIdx.x < n_update) cr_update_body<T>(a, 0, ngroups, (int)blockIdx.x);
  else cr_border_update_body<T>(a, nrt, (int)blockIdx.x - n_update);
}

// The corner: partial[chunk][tile] = sum over the chunk's columns k of Yx[rows of the tile][k] Yx[cols of the tile][k], Yx = rows
// n .. rr of A (the border rows and the right-hand side), k over the columns of every ELIMINATED superblock.  Tiles of 64 x 64 over the lower
// triangle of the (nbr + 1) x nbr corner, four waves = 2 x 2 blocks of 32 x 32.
template <int T>
__global__ __launch_bounds__(256) void cr_border_syrk_kernel(CrArgs a, int kcols, int ntiles, double* __restrict__ part) {
  constexpr int m_ = NBI * T;
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wr = w & 1, wc = w >> 1;
  const int chunk = (int)blockIdx.x / ntiles, tile = (int)blockIdx.x - chunk * ntiles;
  // tile -> (ta >= tb) of the lower triangle, row-major: ta (ta + 1) / 2 + tb
  int ta = 0;
  while ((ta + 1) * (ta + 2) / 2 <= tile) ++ta;
  const int tb = tile - ta * (ta + 1) / 2;
  const int n = a.n, next = a.nbr + 1;
  const size_t lda = (size_t)a.lda;
  const double* const A = a.brd();  // (border rows only)
  const int k0 = chunk * kcols, k1 = k0 + kcols < n ? k0 + kcols : n;
  double4_t acc[2][2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) acc[cb][rb] = (double4_t){0.0, 0.0, 0.0, 0.0};
  int ro[2], co[2];  // extra-row index of this lane for the two row / column blocks, clamped into the matrix
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int r = 64 * ta + 32 * wr + 16 * b + m, c = 64 * tb + 32 * wc + 16 * b + m;
    ro[b] = n + (r < next ? r : next - 1);
    co[b] = n + (c < next ? c : next - 1);
  }
  for (int k = k0; k < k1; k += 16) {
    const int sbk = k / m_;
    if ((sbk & (a.keep - 1)) == 0) continue;  // (uniform) a survivor's columns hold E_j, not Y: they join the dense system
    if (a.nzT != nullptr && (a.nzT[sbk * a.ntr + ta] == 0 || a.nzT[sbk * a.ntr + tb] == 0)) continue;  // Y is zero in the rows of one operand
    double av[4][2], bv[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kk = k + 4 * u + q, kc = kk < k1 ? kk : k1 - 1;
      const bool ok = kk < k1;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        av[u][b] = keep_if(A[(size_t)kc * lda + co[b]], ok);
        bv[u][b] = A[(size_t)kc * lda + ro[b]];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) acc[cb][rb] = mma(av[u][cb], bv[u][rb], acc[cb][rb]);
  }
  double* out = part + ((size_t)chunk * ntiles + tile) * 4096;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(32 * wc + 16 * cb + q + 4 * r) * 64 + 32 * wr + 16 * rb + m] = acc[cb][rb][r];
}

// corner -= the chunks' partial tiles, summed in chunk order (fixed: results do not depend on the launch's scheduling)
template <int T>
__global__ __launch_bounds__(256) void cr_border_syrk_reduce_kernel(CrArgs a, int ntiles, int nchunks, const double* __restrict__ part) {
  const int tile = (int)blockIdx.x >> 4;  // 16 workgroups per tile, one element per thread
  int ta = 0;
  while ((ta + 1) * (ta + 2) / 2 <= tile) ++ta;
  const int tb = tile - ta * (ta + 1) / 2;
  const int n = a.n, next = a.nbr + 1;
  const size_t lda = (size_t)a.lda;
  const int e = (((int)blockIdx.x & 15) << 8) + (int)threadIdx.x;
  const int c = e >> 6, r = e & 63;
  const int row = 64 * ta + r, col = 64 * tb + c;
  if (row >= next || col >= a.nbr || col > row) return;
  const double* p = part + (size_t)tile * 4096 + e;
  const size_t stride = (size_t)ntiles * 4096;
  double sum = 0.0;
  int ch = 0;
  for (; ch + 8 <= nchunks; ch += 8) {  // eight loads in flight, added in chunk order
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(ch + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += v[u];
  }
  for (; ch < nchunks; ++ch) sum += p[(size_t)ch * stride];
  a.brd()[(size_t)(n + col) * lda + n + row] -= sum;
}

// The dense system that is left: the surviving superblocks 0, keep, 2 keep, ... (reduced, not factored; block tridiagonal among
// themselves: D_j, B(j + keep, j), zeros elsewhere -- written, not read: A is only defined where the reduction put something),
// then the border, with the right-hand side as row qn.  qb = band unknowns that survive (only the last survivor can be partial).
// M is qn x qn column-major with pitch ldq; element (r, c), r >= c.
template <int T>
__global__ void cr_border_gather_kernel(CrArgs a, int qb, double* __restrict__ M, int ldq, unsigned* __restrict__ flow_flags, unsigned n_flow,
                                        unsigned* __restrict__ xh_words, unsigned n_xh, int* __restrict__ info_q) {
  constexpr int m_ = NBI * T;
  const int c = (int)blockIdx.x, qn = qb + a.nbr, n = a.n;
  if (c >= qn) {  // the workgroups behind the columns clear what the single-launch solve kernels need (as ba.hip's schur_reduce does)
    const unsigned i0 = (unsigned)(c - qn) * blockDim.x + threadIdx.x, step = (gridDim.x - (unsigned)qn) * blockDim.x;
    if (i0 == 0) *info_q = 0;
    for (unsigned i = i0; i < n_flow; i += step) flow_flags[i] = 0u;
    for (unsigned i = i0; i < n_xh; i += step) xh_words[i] = 0xFFF8BEEFu;
    return;
  }
  const size_t lda = (size_t)a.lda;
  const int keep = a.keep < a.N ? a.keep : a.N;  // (keep >= N: one survivor; the products below stay small)
  const int jc = c < qb ? c / m_ : -1;            // survivor index of the column (-1: a border column)
  const size_t src_col = c < qb ? (size_t)jc * keep * m_ + (size_t)(c - jc * m_) : (size_t)(n + c - qb);
  for (int r = c + (int)threadIdx.x; r <= qn; r += (int)blockDim.x) {
    double v = 0.0;
    if (r < qb) {
      const int jr = r / m_;  // (c < qb here: r >= c)
      if (jr - jc <= 1) v = a.blk(jr * keep, jc * keep)[src_col * lda + (size_t)jr * keep * m_ + (size_t)(r - jr * m_)];  // D_j / B(j + keep, j)
    } else {
      v = a.brd()[src_col * lda + (r < qn ? (size_t)(n + r - qb) : (size_t)a.rr)];
    }
    M[(size_t)c * ldq + r] = v;
  }
}

// The survivors' x and x_c out of the dense solution; t[k] = sum_r x_c[r] Y[r][k] for the columns k of the eliminated superblocks
// (one wave per column)
template <int T>
__global__ __launch_bounds__(256) void cr_border_back_kernel(CrArgs a, int qb, const double* __restrict__ xq, double* __restrict__ tvec,
                                                             const int* __restrict__ info_q) {
  constexpr int m_ = NBI * T;
  const int n = a.n, lane = threadIdx.x & 63;
  const int k = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);  // k < n: a column of the band; n <= k < n + nbr: a border unknown
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // an expired hand-off wait of the dense top's single-launch kernels (info = its n + 1 + block) -> beyond THIS system's size,
    // which is how the callers tell "repeat without the single-launch kernels" from "not positive definite"
    // (a column of the dense system that is not positive definite: reported as a column of the first survivor / the border)
    const int iq = *info_q, qn = qb + a.nbr;
    if (iq != 0 && *a.info == 0) *a.info = iq > qn ? n + a.nbr + 1 + (iq - qn) : (iq <= qb ? iq : n + (iq - qb));
  }
  if (k >= n + a.nbr) return;
  const int sb = k < n ? k / m_ : 0;
  if (k >= n || (sb & (a.keep - 1)) == 0) {  // a survivor or the border: x is final
    const int keep = a.keep < a.N ? a.keep : a.N;
    if (lane == 0) a.x[k] = xq[k < n ? (sb / keep) * m_ + (k - sb * m_) : qb + k - n];
    return;
  }
  if (a.nbr == 0) return;
  const double* col = a.brd() + (size_t)k * a.lda + n;
  double sum = 0.0;
  for (int r = lane; r < a.nbr; r += 64) sum = __builtin_fma(xq[qb + r], col[r], sum);
  sum = wave_sum63(sum);
  if (lane == 63) tvec[k] = sum;
}
// yh_i -= t_i L_i^-1 for every eliminated superblock i >= 1 (L_i^-1[c][c'] = W3_i at c m + c')
template <int T>
__global__ __launch_bounds__(256) void cr_border_yh_kernel(CrArgs a, const double* __restrict__ tvec) {
  constexpr int m_ = NBI * T;
  __shared__ double ts[m_];
  const int i = 1 + (int)blockIdx.x, n = a.n;
  if ((i & (a.keep - 1)) == 0) return;  // (a survivor of the dense top)
  for (int c = threadIdx.x; c < m_; c += 256) ts[c] = i * m_ + c < n ? tvec[i * m_ + c] : 0.0;
  __syncthreads();
  const double* W3 = a.W3 + (size_t)i * ((size_t)m_ * m_);
  for (int cp = threadIdx.x; cp < m_; cp += 256) {
    double sum = 0.0;
    for (int c = 0; c < m_; ++c) sum = __builtin_fma(ts[c], W3[(size_t)c * m_ + cp], sum);  // (the zeros of the triangle are stored)
    a.yh[(size_t)i * m_ + cp] -= sum;
  }
}

// Shape of the dense top of a reduction over N superblocks of m columns (n band unknowns): stride S of the survivors 0, S, 2 S, ...,
// their number and the band unknowns qb they hold (only the last survivor can be partial).  ONE function for the solver and for the
// size of its workspace (gh_arrow_ws_doubles): the two used to disagree on qn when the last survivor is partial or fewer than `top`
// superblocks survive, and the single-launch factorisation's state then lay past the reserved block.
struct TopShape { int S, nsv, qb; };
inline TopShape top_shape(int n, int m, int top) {
  const int N = gh_div_up(n, m);
  int S = 1;
  while (gh_div_up(N, S) > top) S *= 2;
  const int nsv = gh_div_up(N, S);
  const int last = n - (nsv - 1) * S * m;
  return TopShape{S, nsv, (nsv - 1) * m + (last < m ? last : m)};
}

// a launch on `stream` through the context's profiler (GH_LAUNCH times on ctx->stream)
#define CR_LAUNCH_ON(stream_, ...)          \
  do {                                      \
    hipStream_t keep_ = ctx->stream;        \
    ctx->stream = (stream_);                \
    gh_status st_ = [&]() -> gh_status {    \
      GH_LAUNCH(ctx, __VA_ARGS__);          \
      return GH_OK;                         \
    }();                                    \
    ctx->stream = keep_;                    \
    if (st_ != GH_OK) return st_;           \
  } while (0)

// nbr > 0: an arrowhead system (see above) -- A holds n + nbr unknowns, bws the border workspace (gh_arrow_ws_doubles).
template <int T>
gh_status cr_solve_t(gh_ctx* ctx, double* A, int n, int lda, double* dinv, double* W, double* x, int* info_dev, int nbr = 0,
                     double* bws = nullptr, bool allow_flow = true, const uint8_t* border_nz = nullptr, bool compact = false) {
  constexpr int m_ = NBI * T;
  const int N = gh_div_up(n, m_);
  const size_t mm = (size_t)m_ * m_;
  const size_t factor_lds = ((sizeof(Potf2Lds) + 15) & ~(size_t)15) + 16 * 17 * sizeof(double) + (size_t)(T - 1) * NBI * kCrXP * sizeof(double) + 2 * m_ * sizeof(double);
  {
    static bool attr_set[64] = {};
    const int dev = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
    if (!attr_set[dev]) {
      GH_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(cr_factor_kernel<T>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)factor_lds));
      attr_set[dev] = true;
    }
  }
  // Off the critical path, on a side stream, in two launches behind the last level's panels (they overlap the factorisation
  // of the last block): W3 = L_i^-T of every eliminated superblock, then G = W W3^T and yh -- only the backward pass reads
  // them.  One event each way: every marker on the main stream costs its dependent launch chain ~7 us.
  if (!ctx->cr_side && hipStreamCreateWithFlags(&ctx->cr_side, hipStreamNonBlocking) != hipSuccess)
    return gh_set_error(ctx, GH_ERR_HIP, "band solver: side stream");
  while ((int)ctx->cr_events.size() < 2) {
    hipEvent_t ev;
    GH_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    ctx->cr_events.push_back(ev);
  }
  hipStream_t side = ctx->cr_side;
  // workspace: W [2 N m^2], G [2 N m^2], W3 [N m^2], yh [N m]
  CrArgs a{A, lda, n, N, 1, 0, 0, dinv, W, W + 2 * (size_t)N * mm, W + 4 * (size_t)N * mm, W + 5 * (size_t)N * mm, x, info_dev};
  a.nbr = nbr;
  a.rr = n + nbr;
  if (border_nz != nullptr && nbr > 0) {
    a.nbs = gh_div_up(nbr, 16);
    a.ntr = gh_div_up(nbr + 1, 64);
    a.nzY = border_nz;
    a.nzT = border_nz + (size_t)N * a.nbs;
  }
  const int nbs = gh_div_up(nbr, 16), nrt = gh_div_up(nbr, 64);  // 16-row strips / 64-row tiles of the border
  constexpr int NTS = 4 * (T * (T + 1) / 2 + T * T) + m_ / 16, NTE = 4 * (2 * T * T) + m_ / 16;
  auto update_grid = [](int ngroups, int ntask) {  // the mapping of cr_update_kernel
    const int G = ngroups > 4 ? 8 : (ngroups > 2 ? 4 : (ngroups > 1 ? 2 : 1)), kx = 8 / G;
    return 8 * gh_div_up(ntask, kx) * gh_div_up(ngroups, G);
  };
  // DENSE TOP: the reduction stops when at most `top` superblocks survive (0, S, 2 S, ...): the last levels eliminate one or
  // two superblocks each behind a full-length pivot chain (58 us per level at T = 3), while the single-launch dense
  // factorisation takes the block tridiagonal system of four survivors in about the time of ONE level
  const TopShape ts = top_shape(n, m_, (bws != nullptr) ? gh_cr_top(nbr) : 1);
  const int S = ts.S, nsv = ts.nsv;  // stride and number of the survivors
  a.keep = S;
  if (compact) {  // (cr_map.h; lda = gh_cr_compact_lda of the same system)
    int levels = 0;
    while ((1 << levels) < S) ++levels;
    a.map.m = m_;
    a.map.n_band = n;
    a.map.brow = (2 + levels) * m_;
    a.map.levels = levels;
    if (lda < cr_compact_lda(m_, levels, nbr)) return gh_set_error(ctx, GH_ERR_ARG, "band solver: compact columns of %d rows, %d needed", lda, cr_compact_lda(m_, levels, nbr));
  }
  for (int s = 1; s < S; s *= 2) {
    a.s = s;
    a.first = s;
    a.count = (N - s + 2 * s - 1) / (2 * s);
    const int g8 = 8 * gh_div_up(a.count, 8);  // groups of eight eliminated superblocks, one per XCD
    GH_LAUNCH(ctx, "ba_cr_factor", cr_factor_kernel<T>, dim3(a.count), dim3(512), factor_lds, a);
    // (the border strips of an arrowhead system ride in the same launch: they need nothing but L_i either)
    GH_LAUNCH(ctx, "ba_cr_panels", cr_panels_kernel<T>, dim3(g8 * (8 * T + 1 + nbs)), dim3(256), 0, a, 0, 8 * T + 1 + nbs);
    if (2 * s >= S) {  // the last level: everything the side work reads is (or will be, in stream order) complete here
      GH_HIP(ctx, hipEventRecord(ctx->cr_events[0], ctx->stream));
      GH_HIP(ctx, hipStreamWaitEvent(side, ctx->cr_events[0], 0));
      CrArgs b = a;
      b.s = 0;
      b.first = 1;
      b.count = N - 1;
      const int ge = 8 * gh_div_up(N - 1, 8);
      CR_LAUNCH_ON(side, "ba_cr_inverse", cr_panels_kernel<T>, dim3(ge * 4 * T), dim3(256), 0, b, 1, 4 * T);
      CR_LAUNCH_ON(side, "ba_cr_backprep", cr_update_kernel<T>, dim3(update_grid(N - 1, NTE)), dim3(256), 0, b, 1, N - 1);
      GH_HIP(ctx, hipEventRecord(ctx->cr_events[1], side));
    }
    const int nsurv = gh_div_up(N, 2 * s);
    if (nbr > 0) {
      const int nu = update_grid(nsurv, NTS);
      GH_LAUNCH(ctx, "ba_cr_update", cr_update_both_kernel<T>, dim3(nu + nsurv * nrt * T * 4), dim3(256), 0, a, nsurv, nu, nrt);
    } else {
      GH_LAUNCH(ctx, "ba_cr_update", cr_update_kernel<T>, dim3(update_grid(nsurv, NTS)), dim3(256), 0, a, 0, nsurv);
    }
  }
  // what is left: block 0 with no neighbours (stride >= N), or the dense top
  a.s = S;
  a.first = 0;
  a.count = 1;
  if (nbr > 0 || nsv > 1) {
    // the survivors are not factored on their own: they join the border in the dense system that is left
    const int qb = ts.qb;
    const int qn = qb + nbr, ldq = (qn + 1 + 15) & ~15;
    const int ntr = gh_div_up(nbr + 1, 64), ntiles = ntr * (ntr + 1) / 2;
    const BorderChunks bc = border_chunks(n, m_, nbr);
    const int nchunks = bc.n;
    double* part = bws;
    double* Mq = part + (size_t)nchunks * ntiles * 4096;
    double* dinv_q = Mq + (size_t)ldq * qn;
    double* xwork_q = dinv_q + (size_t)gh_div_up(qn, NBI) * (NBI * NBI);
    double* work_q = xwork_q + (size_t)2 * 64 * (qn + 1);
    double* xq = work_q + (((size_t)qn + 15) & ~(size_t)15);
    double* xh_q = xq + (((size_t)qn + 15) & ~(size_t)15);
    double* tvec = xh_q + (size_t)gh_div_up(qn, NBI) * NBI;
    // the corner through the single-launch factorisation when its shape fits (the caller holds gh_potrf_flow_mutex)
    const size_t flow_words = allow_flow ? gh_potrf_flow_words(ctx, qn, 1) : 0;
    // the dense top reports into a word of its own (the reduction's factor kernels use *info_dev with THEIR column numbers)
    int* info_q = reinterpret_cast<int*>(tvec + (((size_t)n + 15) & ~(size_t)15));
    unsigned* flow_q = flow_words ? reinterpret_cast<unsigned*>(tvec + (((size_t)n + 15) & ~(size_t)15) + 16) : nullptr;
    const unsigned n_flow = flow_words ? (unsigned)gh_potrf_flow_flag_words(qn, 1) : 0u;
    const unsigned n_xh = (unsigned)gh_div_up(qn, NBI) * NBI * 2u;
    if (nchunks > 0) {
      GH_LAUNCH(ctx, "ba_cr_border_syrk", cr_border_syrk_kernel<T>, dim3(nchunks * ntiles), dim3(256), 0, a, bc.kc, ntiles, part);
      GH_LAUNCH(ctx, "ba_cr_border_reduce", cr_border_syrk_reduce_kernel<T>, dim3(16 * ntiles), dim3(256), 0, a, ntiles, nchunks,
                (const double*)part);
    }
    GH_LAUNCH(ctx, "ba_cr_border_gather", cr_border_gather_kernel<T>, dim3(qn + 8), dim3(256), 0, a, qb, Mq, ldq, flow_q, n_flow,
              reinterpret_cast<unsigned*>(xh_q), n_xh, info_q);
    GH_TRY(gh_potrf_dev_impl(ctx, Mq, qn, ldq, info_q, 1, dinv_q, xwork_q, flow_q, false, true));
    GH_TRY(gh_potrs_bwd_dev_impl(ctx, Mq, qn, ldq, xq, work_q, dinv_q, Mq + qn, ldq, allow_flow ? xh_q : nullptr, info_q, true));
    if (S > 1) GH_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->cr_events[1], 0));
    GH_LAUNCH(ctx, "ba_cr_border_back", cr_border_back_kernel<T>, dim3(gh_div_up(n + nbr, 4)), dim3(256), 0, a, qb, (const double*)xq, tvec, (const int*)info_q);
    if (S > 1 && nbr > 0) GH_LAUNCH(ctx, "ba_cr_border_yh", cr_border_yh_kernel<T>, dim3(N - 1), dim3(256), 0, a, (const double*)tvec);
  } else {
  GH_LAUNCH(ctx, "ba_cr_factor", cr_factor_kernel<T>, dim3(1), dim3(512), factor_lds, a);  // (also solves: x_0 is final)
  if (S > 1) GH_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->cr_events[1], 0));
  }
  for (int s = S / 2; s >= 1; s /= 2) {
    a.s = s;
    a.first = s;
    a.count = (N - s + 2 * s - 1) / (2 * s);
    GH_LAUNCH(ctx, "ba_cr_back", cr_back_kernel<T>, dim3(8 * gh_div_up(a.count, 8) * (m_ / 16)), dim3(256), 0, a);
  }
  return GH_OK;
}

}  // namespace

// Superblocks the dense top takes: GSLAM_HIP_CR_TOP = 1 .. 16 (1 = reduce down to superblock 0, the form before the dense top);
// default 2 for a band and 4 with a border, whose dense corner is there anyway.  Measured at C4, LM iterations per second on the
// resident graph by top = 1 / 2 / 4 / 8: band 1717 / 1803 / 1747 / 1449, + 20 loop closures 1234 / 1320 / 1335 / 1231
// (profiles/ba_dense_top_r05.txt): the single-launch factorisation costs ~16 us per 64 columns, a level ~85 us.
int gh_cr_top_env() {
  static const int v = [] {
    const char* e = getenv("GSLAM_HIP_CR_TOP");
    const int t = e ? atoi(e) : 0;
    return t < 1 ? 0 : (t > 16 ? 16 : t);
  }();
  return v;
}
int gh_cr_top(int nbr) { return gh_cr_top_env() ? gh_cr_top_env() : (nbr > 0 ? 4 : 2); }

// Rows of a column of the COMPACT layout (cr_map.h) for this system as cr_solve_t will reduce it; *brow = local row of the first
// border row / of the right-hand side when there is no border.  Only systems solved WITH the border workspace (dense top).
int gh_cr_compact_lda(int n_band, int T, int nbr, int* brow) {
  const int m = NBI * T;
  const TopShape ts = top_shape(n_band, m, gh_cr_top(nbr));
  int levels = 0;
  while ((1 << levels) < ts.S) ++levels;
  if (brow) *brow = (2 + levels) * m;
  return cr_compact_lda(m, levels, nbr);
}

// border workspace of an arrowhead solve (doubles): the corner update's partial tiles, the dense system of superblock 0 + border
// and what chol.hip's dense path needs for it, the backward pass's t vector
size_t gh_arrow_ws_doubles(const gh_ctx* ctx, int n_band, int T, int nbr) {
  // (qn of THIS system, as cr_solve_t computes it: gh_potrf_flow_words is not monotonic in qn -- it is 0 once the shape exceeds the
  //  CU count -- so a bound from the largest possible qn may reserve nothing where the real qn needs megabytes)
  const size_t m = (size_t)NBI * T, qn = (size_t)top_shape(n_band, NBI * T, gh_cr_top(nbr)).qb + (size_t)nbr, ldq = (qn + 1 + 15) & ~(size_t)15;
  const size_t ntr = ((size_t)nbr + 1 + 63) / 64, ntiles = ntr * (ntr + 1) / 2;
  const size_t nchunks = (size_t)border_chunks(n_band, (int)m, nbr).n;
  const size_t qb = (qn + NBI - 1) / NBI;
  return nchunks * ntiles * 4096 + ldq * qn + qb * (NBI * NBI) + 2 * 64 * (qn + 1) + 2 * ((qn + 15) & ~(size_t)15) + qb * NBI +
         (((size_t)n_band + 15) & ~(size_t)15) + gh_potrf_flow_words(ctx, (int)qn, 1) / 2 + 64;
}

// Structure of the border through the reduction (host, once per topology).  init[i * nbs + strip] != 0 iff the 16-row strip
// of the border rows has a structural non-zero among the band columns of superblock i (a border camera and a band camera that
// see a common point).  Eliminating i at level s fills the strips of i into i - s and i + s (E_u -= Y_i W_u^T); out receives
// nzY [N][nbs] -- the strips of E_i when i is eliminated, all set for the survivors of the dense top -- then nzT [N][ntr]: per
// 64-row tile of the corner, with the tile that holds the right-hand-side row (it rides along as row nbr) always set.
// Conservative by construction: a block marked zero is never written by any kernel and was cleared by the assembly.
size_t gh_cr_border_symbolic_bytes(int n_band, int T, int nbr) {
  const size_t N = (size_t)gh_div_up(n_band, NBI * T);
  return N * ((size_t)gh_div_up(nbr, 16) + (size_t)gh_div_up(nbr + 1, 64));
}
void gh_cr_border_symbolic(int n_band, int T, int nbr, const uint8_t* init, uint8_t* out) {
  const int m = NBI * T, N = gh_div_up(n_band, m), nbs = gh_div_up(nbr, 16), ntr = gh_div_up(nbr + 1, 64);
  std::vector<uint8_t> cur(init, init + (size_t)N * nbs);
  uint8_t* nzY = out;
  uint8_t* nzT = out + (size_t)N * nbs;
  memset(nzY, 1, (size_t)N * nbs);
  const int top = gh_cr_top(nbr);
  int S = 1;
  while (gh_div_up(N, S) > top) S *= 2;
  for (int s = 1; s < S; s *= 2)
    for (int i = s; i < N; i += 2 * s) {
      memcpy(nzY + (size_t)i * nbs, &cur[(size_t)i * nbs], (size_t)nbs);
      for (int side = -1; side <= 1; side += 2) {
        const int j = i + side * s;
        if (j < 0 || j >= N) continue;
        for (int t = 0; t < nbs; ++t) cur[(size_t)j * nbs + t] |= cur[(size_t)i * nbs + t];
      }
    }
  for (int i = 0; i < N; ++i)
    for (int t = 0; t < ntr; ++t) {
      uint8_t v = (t == nbr / 64) ? 1 : 0;
      for (int w = 0; w < 4; ++w)
        if (4 * t + w < nbs) v |= nzY[(size_t)i * nbs + 4 * t + w];
      nzT[(size_t)i * ntr + t] = v;
    }
}

// C-ABI face of the two functions above for tests and tools (host only, no GPU): returns the bytes `out` needs (N * (nbs + ntr))
// and fills it when out != NULL and out_bytes suffices; 0 = bad arguments.
extern "C" size_t gh_cr_border_structure(int n_band, int tiles, int nbr, const uint8_t* init, uint8_t* out, size_t out_bytes) {
  if (n_band < 1 || tiles < 1 || tiles > 3 || nbr < 1) return 0;
  const size_t need = gh_cr_border_symbolic_bytes(n_band, tiles, nbr);
  if (out != nullptr && init != nullptr && out_bytes >= need) gh_cr_border_symbolic(n_band, tiles, nbr, init, out);
  return need;
}

// Tiles per superblock for a half-bandwidth of `hbw` scalars (A[r][c] = 0 for r - c > hbw), 0 = the band is too wide for
// this solver (or the matrix too small to gain from it): the caller stays on the dense factorisation.
int gh_cr_tiles(int n, int hbw) {
  if (hbw < 0 || n < 1) return 0;
  const int T = hbw <= NBI ? 1 : (hbw <= 2 * NBI ? 2 : (hbw <= 3 * NBI ? 3 : 0));
  if (T == 0) return 0;
  if (gh_div_up(n, NBI * T) < 4) return 0;
  return T;
}
size_t gh_cr_dinv_doubles(int n, int T) { return (size_t)gh_div_up(n, NBI * T) * T * (NBI * NBI); }
// workspace of the panels: W and G (two sides each), W3, yh
size_t gh_cr_panel_doubles(int n, int T) {
  const size_t N = (size_t)gh_div_up(n, NBI * T), m = (size_t)NBI * T;
  return N * (5 * m * m + m);
}

// Solve A x = b for a band matrix: A n x n column-major lower triangle (overwritten), b in row n of A (lda > n), T from
// gh_cr_tiles; everything outside the band must be ZERO in the lower triangle (the fill lands there).  x_dev: n doubles.
// *info_dev: 0 or the first column + 1 of a diagonal tile that is not positive definite.  Asynchronous on the ctx stream.
gh_status gh_cr_solve_dev_impl(gh_ctx* ctx, double* A, int n, int lda, int T, double* dinv, double* W, double* x_dev,
                               int* info_dev, bool info_ready) {
  // (no border workspace: the reduction runs down to superblock 0 -- the form of rounds 4 / 5a, kept for A/B runs and tests)
  if (!info_ready) GH_HIP(ctx, hipMemsetAsync(info_dev, 0, sizeof(int), ctx->stream));
  switch (T) {
    case 1: return cr_solve_t<1>(ctx, A, n, lda, dinv, W, x_dev, info_dev);
    case 2: return cr_solve_t<2>(ctx, A, n, lda, dinv, W, x_dev, info_dev);
    case 3: return cr_solve_t<3>(ctx, A, n, lda, dinv, W, x_dev, info_dev);
    default: return gh_set_error(ctx, GH_ERR_ARG, "gh_cr_solve: %d tiles per superblock", T);
  }
}

// Arrowhead solve: A holds n_band + nbr unknowns (band first, border last) and the right-hand side in row n_band + nbr; the
// lower triangle of the band part must be zero outside the band, the border rows are dense.  x_dev: n_band + nbr doubles.
// border_nz (device, may be null = E dense): the structure gh_cr_border_symbolic computed for this system, N * (nbs + ntr) bytes.
// allow_flow = false: the dense top stays off chol.hip's single-launch kernels (the caller saw one of their bounded waits expire:
// *info_dev > n_band + nbr).
gh_status gh_arrow_solve_dev_impl(gh_ctx* ctx, double* A, int n_band, int nbr, int lda, int T, double* dinv, double* W, double* bws,
                                  double* x_dev, int* info_dev, bool info_ready, bool allow_flow, const uint8_t* border_nz, bool compact) {
  if (!info_ready) GH_HIP(ctx, hipMemsetAsync(info_dev, 0, sizeof(int), ctx->stream));
  switch (T) {
    case 1: return cr_solve_t<1>(ctx, A, n_band, lda, dinv, W, x_dev, info_dev, nbr, bws, allow_flow, border_nz, compact);
    case 2: return cr_solve_t<2>(ctx, A, n_band, lda, dinv, W, x_dev, info_dev, nbr, bws, allow_flow, border_nz, compact);
    case 3: return cr_solve_t<3>(ctx, A, n_band, lda, dinv, W, x_dev, info_dev, nbr, bws, allow_flow, border_nz, compact);
    default: return gh_set_error(ctx, GH_ERR_ARG, "gh_arrow_solve: %d tiles per superblock", T);
  }
}

namespace {
__global__ void cr_rhs_to_row_kernel(const double* __restrict__ rhs, double* __restrict__ A, int lda, int row, int n) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < n) A[(size_t)j * lda + row] = rhs[j];
}
}  // namespace

extern "C" gh_status gh_band_solve_dev(gh_ctx* ctx, double* A_dev, int n, int lda, int half_bandwidth, double* b_dev,
                                       int* info) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, A_dev && b_dev && info && n > 0 && lda > n && half_bandwidth >= 0);
  const int T = gh_cr_tiles(n, half_bandwidth);
  if (T == 0)
    return gh_set_error(ctx, GH_ERR_ARG, "gh_band_solve_dev: half-bandwidth %d of n = %d does not fit (<= %d, >= 4 superblocks)",
                        half_bandwidth, n, 3 * NBI);
  void* scratch = nullptr;
  const size_t nd = gh_cr_dinv_doubles(n, T), nw = gh_cr_panel_doubles(n, T), nbw = gh_arrow_ws_doubles(ctx, n, T, 0);
  std::lock_guard<std::mutex> flow_lock(gh_potrf_flow_mutex(ctx->device));  // (the dense top may run as the single-launch factorisation)
  GH_TRY(gh_scratch(ctx, 256 + (nd + nw + nbw + (size_t)n) * sizeof(double), &scratch));
  int* info_dev = (int*)scratch;
  double* dinv = (double*)((char*)scratch + 256);
  double* W = dinv + nd;
  double* bws = W + nw;
  double* x = bws + nbw;
  GH_LAUNCH(ctx, "ba_rhs_row", cr_rhs_to_row_kernel, dim3(gh_div_up(n, 256)), dim3(256), 0, (const double*)b_dev, A_dev, lda, n, n);
  GH_TRY(gh_arrow_solve_dev_impl(ctx, A_dev, n, 0, lda, T, dinv, W, bws, x, info_dev, false, true, nullptr, false));
  GH_HIP(ctx, hipMemcpyAsync(b_dev, x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  GH_HIP(ctx, hipMemcpyAsync(info, info_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GH_OK;
}

// Arrowhead SPD solve (test / tool entry, like gh_band_solve_dev): A_dev n x n column-major lower triangle (lda > n, overwritten),
// the first n_band unknowns form a band of `half_bandwidth`, the last n - n_band are the dense border.
extern "C" gh_status gh_arrow_solve_dev(gh_ctx* ctx, double* A_dev, int n, int lda, int n_band, int half_bandwidth, double* b_dev,
                                        int* info) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, A_dev && b_dev && info && n > 0 && lda > n && half_bandwidth >= 0 && n_band > 0 && n_band < n);
  const int T = gh_cr_tiles(n_band, half_bandwidth), nbr = n - n_band;
  if (T == 0)
    return gh_set_error(ctx, GH_ERR_ARG, "gh_arrow_solve_dev: half-bandwidth %d of n_band = %d does not fit (<= %d, >= 4 superblocks)",
                        half_bandwidth, n_band, 3 * NBI);
  void* scratch = nullptr;
  const size_t nd = gh_cr_dinv_doubles(n_band, T), nw = gh_cr_panel_doubles(n_band, T), nbw = gh_arrow_ws_doubles(ctx, n_band, T, nbr);
  std::lock_guard<std::mutex> flow_lock(gh_potrf_flow_mutex(ctx->device));  // (the dense corner may run as the single-launch factorisation)
  GH_TRY(gh_scratch(ctx, 256 + (nd + nw + nbw + (size_t)n) * sizeof(double), &scratch));
  int* info_dev = (int*)scratch;
  double* dinv = (double*)((char*)scratch + 256);
  double* W = dinv + nd;
  double* bws = W + nw;
  double* x = bws + nbw;
  GH_LAUNCH(ctx, "ba_rhs_row", cr_rhs_to_row_kernel, dim3(gh_div_up(n, 256)), dim3(256), 0, (const double*)b_dev, A_dev, lda, n, n);
  GH_TRY(gh_arrow_solve_dev_impl(ctx, A_dev, n_band, nbr, lda, T, dinv, W, bws, x, info_dev, false, true, nullptr, false));
  GH_HIP(ctx, hipMemcpyAsync(b_dev, x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  GH_HIP(ctx, hipMemcpyAsync(info, info_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GH_OK;
}

// ---------------------------------------------------------------- the compact layout through the C ABI (tests, tools)
// Layout of the compact columns (cr_map.h) gh_ba_solve keeps its reduced camera system in, for a system of n_band band unknowns of
// `half_bandwidth` + nbr border unknowns: *lda = rows per column, *m = superblock columns, *brow = local row of the first border row
// (the right-hand side is local row brow + nbr).  Element (r, c), r >= c, lives at c * lda + local(r, c):
//   r >= n_band                  brow + (r - n_band)
//   I - J <= 1 (I = r / m, J = c / m)   r - J m
//   I - J = 2^k, k >= 1          (1 + k) m + (r - I m)       (fill of the reduction: zero on entry)
// Returns 0 when the band does not fit the solver (gh_band_solve_dev's rule).
extern "C" int gh_cr_compact_layout(int n_band, int half_bandwidth, int nbr, int* lda, int* m, int* brow) {
  const int T = gh_cr_tiles(n_band, half_bandwidth);
  if (T == 0 || nbr < 0) return 0;
  int br = 0;
  const int l = gh_cr_compact_lda(n_band, T, nbr, &br);
  if (lda) *lda = l;
  if (m) *m = NBI * T;
  if (brow) *brow = br;
  return 1;
}

// gh_arrow_solve_dev / gh_band_solve_dev (n_band == n) on a matrix in the compact layout: A_dev holds n columns of `lda` doubles
// (gh_cr_compact_layout), right-hand side b_dev (n doubles, overwritten by x).  Test / tool entry.
extern "C" gh_status gh_arrow_solve_compact_dev(gh_ctx* ctx, double* A_dev, int n, int lda, int n_band, int half_bandwidth,
                                                double* b_dev, int* info) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, A_dev && b_dev && info && n > 0 && half_bandwidth >= 0 && n_band > 0 && n_band <= n);
  const int T = gh_cr_tiles(n_band, half_bandwidth), nbr = n - n_band;
  if (T == 0)
    return gh_set_error(ctx, GH_ERR_ARG, "gh_arrow_solve_compact_dev: half-bandwidth %d of n_band = %d does not fit (<= %d, >= 4 superblocks)",
                        half_bandwidth, n_band, 3 * NBI);
  int brow = 0;
  GH_CHECK_ARG(ctx, lda >= gh_cr_compact_lda(n_band, T, nbr, &brow));
  void* scratch = nullptr;
  const size_t nd = gh_cr_dinv_doubles(n_band, T), nw = gh_cr_panel_doubles(n_band, T), nbw = gh_arrow_ws_doubles(ctx, n_band, T, nbr);
  std::lock_guard<std::mutex> flow_lock(gh_potrf_flow_mutex(ctx->device));
  GH_TRY(gh_scratch(ctx, 256 + (nd + nw + nbw + (size_t)n) * sizeof(double), &scratch));
  int* info_dev = (int*)scratch;
  double* dinv = (double*)((char*)scratch + 256);
  double* W = dinv + nd;
  double* bws = W + nw;
  double* x = bws + nbw;
  // the right-hand side into its row: local row brow + nbr of every column
  GH_LAUNCH(ctx, "ba_rhs_row", cr_rhs_to_row_kernel, dim3(gh_div_up(n, 256)), dim3(256), 0, (const double*)b_dev, A_dev, lda, brow + nbr, n);
  GH_TRY(gh_arrow_solve_dev_impl(ctx, A_dev, n_band, nbr, lda, T, dinv, W, bws, x, info_dev, false, true, nullptr, true));
  GH_HIP(ctx, hipMemcpyAsync(b_dev, x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  GH_HIP(ctx, hipMemcpyAsync(info, info_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GH_OK;
}
