// The 64x64 diagonal-block code shared by the dense factorisation (chol.hip) and the cyclic-reduction solver of banded
// reduced camera systems (chol_cr.hip): potf2 + inverse of a block held in LDS.  Included INSIDE the anonymous namespace of
// the including file, which must have defined NBI (= 64), double4_t and lds_barrier() before.
#ifndef CHOL_STAMP
#define CHOL_STAMP(i) \
  do {                \
  } while (0)
#endif
#ifndef CHAIN_STAMP
#define CHAIN_STAMP(i) \
  do {                 \
  } while (0)
#define CHAIN_SINCE(i, from) \
  do {                       \
  } while (0)
#endif
#ifndef FLOW_SKIP
#define FLOW_SKIP(w, bit) false
#endif

// ---------------------------------------------------------------- constants of the 64x64 diagonal-block code
constexpr int NBS = 16;      // sub-step width inside a 64-block
constexpr int LP = NBI + 1;  // LDS pitch (column-major: element (r, c) at c * LP + r)

// ---------------------------------------------------------------- potf2 + inverse of a 64x64 diagonal block
// 1 / sqrt(d) to ~1 ulp: hardware estimate r0 (v_rsq_f64: relative error <= ~2^-24), then ONE correction
// r0 (1 + e/2 + 3 e^2 / 8) with e = 1 - d r0^2: the next term, 5 e^3 / 16 <= 2^-70, is far below an ulp.  Written with
// q = e / 2 so that every constant is an inline operand (0.5, 1.0): r0 + r0 (q + 1.5 q^2), 1.5 q = q + q / 2 -- a 0.375
// costs two v_mov per call to build, on the one wave that holds the pivots.  d > 0 is checked by the caller
__device__ __forceinline__ double rsqrt_nr(double d) {
  const double r0 = __builtin_amdgcn_rsq(d);
  const double e = __builtin_fma(-(d * r0), r0, 1.0);
  const double q = e * 0.5;
  const double b = __builtin_fma(__builtin_fma(q, 0.5, q), q, q);
  return __builtin_fma(r0, b, r0);
}

// value of lane `l` (wave-uniform index) broadcast through SGPRs
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// Right-looking elimination of a 16-column panel held one ROW per lane (lane = row index inside the 64-block,
// s[c] = column k + c): factors the 16x16 diagonal block and solves the rows below it in the same pass.
// Column J, once scaled, goes to its final place in LDS; the multiplier of the NEXT pivot column comes through
// v_readlane (it is on the critical chain), the others are same-address LDS reads (broadcast), which have a
// whole pivot step of slack -- 2 readlanes + ~8 LDS instructions per column instead of 30 readlanes.
template <int J, int C>
__device__ __forceinline__ void panel16_update(double (&s)[NBS], double l, const double* colJ) {
  if constexpr (C < NBS) {
    s[C] = __builtin_fma(-l, colJ[C], s[C]);  // A[i][k+C] -= L[i][k+J] L[k+C][k+J]
    panel16_update<J, C + 1>(s, l, colJ);
  }
}
// PUB (k >= 1 only): the wave also leaves 1/sqrt(d) in row 0 of the column -- a slot of the upper triangle, which nobody
// reads as part of the matrix -- and, after every pivot, the number of finished columns in `*progress` (LDS executes a
// wave's instructions in order, so whoever sees the count sees the column): another wave can follow the panel
// (potf2_chain_lds)
template <int J, bool PUB = false>
__device__ __forceinline__ void panel16_factor(double (&s)[NBS], double (&rinv)[NBS], double* As, int k, int lane,
                                               bool& bad, int* progress) {
  if constexpr (J < NBS) {
    const double d = readlane_f64(s[J], k + J);
    if (!(d > 0.0)) bad = true;
    const double rs = rsqrt_nr(d);
    rinv[J] = rs;
    double l = (lane >= k + J) ? s[J] * rs : 0.0;  // the pivot lane holds d itself: d * rs = sqrt(d)
    double* col = As + (k + J) * LP;
    if constexpr (PUB) col[lane] = lane == 0 ? rs : l;
    else col[lane] = l;
    if constexpr (J + 1 < NBS) s[J + 1] = __builtin_fma(-l, readlane_f64(l, k + J + 1), s[J + 1]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if constexpr (PUB) {
      if (lane == 0) __hip_atomic_store(progress, J + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    panel16_update<J, J + 2>(s, l, col + k);
    panel16_factor<J + 1, PUB>(s, rinv, As, k, lane, bad, progress);
  }
}
template <int J>
__device__ __forceinline__ void panel16_factor(double (&s)[NBS], double (&rinv)[NBS], double* As, int k, int lane,
                                               bool& bad) {
  panel16_factor<J, false>(s, rinv, As, k, lane, bad, reinterpret_cast<int*>(As));
}

// (Measured on the box, tools/lat_probe.hip + tools/chol_probe.hip: ONE wave issues a v_fma_f64 every ~7 cycles whether or
// not it depends on the previous one, v_rsq_f64 takes 18, a v_readlane pair feeding a VALU operand 23-31.  The ~45
// instructions of a pivot step therefore cost ~300 cycles however they are ordered: deferring the LDS-fed updates by one
// or two steps so that nothing on the chain waits for LDS gave 310 cycles per pivot against 298.  The panel is bound by
// the instruction count of the single wave that holds it.)
// M = L^-1 of a 16x16 block, column `i` per lane (all four 16-lane rows redundantly): forward substitution on e_i,
// right-looking, with the (lane-uniform) entries of L fetched by same-address LDS reads.  Lcol(t) -> &L[0][t].
template <int T, int R>
__device__ __forceinline__ void inv16_update(double (&acc)[NBS], double mt, const double* lcol) {
  if constexpr (R < NBS) {
    acc[R] = __builtin_fma(lcol[R], mt, acc[R]);  // acc[R] += L[R][T] M[T][i]
    inv16_update<T, R + 1>(acc, mt, lcol);
  }
}
template <int T>
__device__ __forceinline__ void inv16(double (&acc)[NBS], double (&mi)[NBS], const double (&rinv)[NBS], const double* L16,
                                      int i) {
  if constexpr (T < NBS) {
    mi[T] = ((T == i) ? 1.0 : -acc[T]) * rinv[T];  // acc[T] is still zero for T < i
    inv16_update<T, T + 1>(acc, mi[T], L16 + T * LP);
    inv16<T + 1>(acc, mi, rinv, L16, i);
  }
}

// the same with a small register footprint (a wave of a 512-thread workgroup has 256 registers): 1/sqrt(d) read when it
// is needed, every entry of M stored as soon as it exists (column i of the block at `mcol`, by the lanes `store`)
template <int T>
__device__ __forceinline__ void inv16_lean(double (&acc)[NBS], const double* rinv_lds, const double* L16, double* mcol, int i,
                                           bool store) {
  if constexpr (T < NBS) {
    const double mt = ((T == i) ? 1.0 : -acc[T]) * rinv_lds[T];
    if (store) mcol[T] = mt;
    inv16_update<T, T + 1>(acc, mt, L16 + T * LP);
    inv16_lean<T + 1>(acc, rinv_lds, L16, mcol, i, store);
  }
}
// the same, following a panel that is still being factored by another wave (panel16_factor<.., PUB>), four columns at a
// time (the LDS reads of a group go out together; the last group is the cheapest: 6 of the 120 updates)
template <int T>
__device__ __forceinline__ void inv16_chase(double (&acc)[NBS], const double* L16, const double* rs_row, int* progress,
                                            double* mcol, int i, bool store) {
  if constexpr (T < NBS) {
    if constexpr (T % 4 == 0) {
      while (__hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < T + 4) {
      }
      asm volatile("" ::: "memory");
    }
    const double mt = ((T == i) ? 1.0 : -acc[T]) * rs_row[T * LP];
    if (store) mcol[T] = mt;
    inv16_update<T, T + 1>(acc, mt, L16 + T * LP);
    inv16_chase<T + 1>(acc, L16, rs_row, progress, mcol, i, store);
  }
}

// One 16x16 output tile of a small LDS-resident product on v_mfma_f64_16x16x4_f64:
//   D[i][j] = sum_{t < 4 KS} a(i, t) * b(t, j);  lane supplies a(lane & 15, 4 ks + lane >> 4) and b(4 ks + lane >> 4, lane & 15),
//   and receives D[(lane >> 4) + 4 r][lane & 15] in acc[r].
template <int KS, typename FA, typename FB>
__device__ __forceinline__ double4_t lds_mma(FA a, FB b, int lane) {
  double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
  const int m = lane & 15, q = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a(m, 4 * ks + q), b(4 * ks + q, m), acc, 0, 0, 0);
  return acc;
}

struct Potf2Lds {
  double As[NBI * LP];  // L  (r, c) at c * LP + r  (only the lower triangle is meaningful)
  double Ms[NBI * LP];  // M = L^-1, same layout
  double Ts[32 * 33];
  double rinv[NBI];
  int bad;
  // potf2_chain_lds: wave 0 is through its last pivot / what the side wave found out meanwhile / columns of the last
  // panel that are final / the off-diagonal block of the first 32x32 inverse is in place
  int pivots_done, next_ready, progress, x10_done;
};

// Factor the 64x64 block held in sh.As (lower, in place) and build M = L^-1 in sh.Ms.  Called by all 256 threads;
// sh.As must be complete (identity padding for missing rows/cols), sh.bad cleared, and a barrier passed.
// (two halves, so that a caller can slip other work -- a prefetch -- between the factorisation and the inversion)
// `idle`: run once by the waves 1..3 while wave 0 is busy with the first 16 pivots
// NW = waves of the calling workgroup (4 or 8): the pivots and the inversions belong to the waves 0..3, the MFMA tiles of
// the trailing updates are dealt over all of them, every wave passes every barrier
// (`tid`: the caller's thread index; the dataflow kernel passes an opaque copy per phase so that the address arithmetic of
// one phase is not kept alive in registers -- or scratch -- through all the others)
template <int NW = 4, typename F>
__device__ __forceinline__ void potf2_factor_lds(Potf2Lds& sh, F&& idle, int tid = threadIdx.x) {
  double* As = sh.As;
  const int lane = tid & 63, wv = NW == 4 ? tid >> 6 : __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, q = lane >> 4;
  for (int k = 0; k < NBI; k += NBS) {
    // (a) 16-column panel: diagonal block + rows below in one right-looking pass, wave 0, lane = row
    if (wv != 0 && k == 0) idle();
    if (wv == 0) {
      double s[NBS], ri[NBS];
#pragma unroll
      for (int c = 0; c < NBS; ++c) s[c] = As[(k + c) * LP + lane];
      bool bad = false;
      panel16_factor<0>(s, ri, As, k, lane, bad);
      if (lane == 0) {
#pragma unroll
        for (int c = 0; c < NBS; ++c) sh.rinv[k + c] = ri[c];
      }
      if (bad) sh.bad = 1;
    }
    lds_barrier();
    CHOL_STAMP(2 + k / 8);
    // (b) rank-16 update of the trailing lower triangle on MFMA: tiles (ti >= tj), round-robin over the waves
    const int nb = (NBI - (k + NBS)) / NBS;
    for (int t = wv; t < nb * (nb + 1) / 2; t += NW) {
      const int ti = t < 1 ? 0 : (t < 3 ? 1 : 2), tj = t - ti * (ti + 1) / 2;  // nb <= 3
      const int rb = k + NBS + 16 * ti, cb = k + NBS + 16 * tj;
      const double4_t u = lds_mma<4>([&](int i, int tt) { return As[(k + tt) * LP + rb + i]; },
                                     [&](int tt, int j) { return As[(k + tt) * LP + cb + j]; }, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) As[(cb + m) * LP + rb + q + 4 * r] -= u[r];
    }
    lds_barrier();
    CHOL_STAMP(3 + k / 8);
  }
}

template <int NW = 4>
__device__ __forceinline__ void potf2_invert_lds(Potf2Lds& sh, int tid = threadIdx.x) {
  double* As = sh.As;
  double* Ms = sh.Ms;
  double* Ts = sh.Ts;
  const int lane = tid & 63, wv = NW == 4 ? tid >> 6 : __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, q = lane >> 4;
  // inverses of the four 16x16 diagonal blocks, one per wave (column m of M per lane)
  if (NW == 4 || wv < 4) {
    const int b0 = 16 * wv;
    double acc[NBS], ri[NBS], mi[NBS];
#pragma unroll
    for (int j = 0; j < NBS; ++j) {
      acc[j] = 0.0;
      ri[j] = sh.rinv[b0 + j];
    }
    inv16<0>(acc, mi, ri, As + b0 * LP + b0, m);
    if (lane < NBS) {
#pragma unroll
      for (int j = 0; j < NBS; ++j) Ms[(b0 + m) * LP + b0 + j] = mi[j];
    }
  }
  lds_barrier();
  CHOL_STAMP(10);
  // block-recursive inverse: inv([A 0; C B]) = [A^-1 0; -B^-1 C A^-1  B^-1]
  // level 32: two independent pairs of 16x16 blocks (waves 0 and 1)
  if (wv < 2) {
    const int b0 = 32 * wv;
    const double4_t t1 = lds_mma<4>([&](int i, int t) { return As[(b0 + t) * LP + b0 + 16 + i]; },   // C[i][t]
                                    [&](int t, int j) { return Ms[(b0 + j) * LP + b0 + t]; }, lane);  // A^-1[t][j]
#pragma unroll
    for (int r = 0; r < 4; ++r) Ts[(16 * wv + m) * 33 + q + 4 * r] = t1[r];
  }
  lds_barrier();
  if (wv < 2) {
    const int b0 = 32 * wv;
    const double4_t x = lds_mma<4>([&](int i, int t) { return Ms[(b0 + 16 + t) * LP + b0 + 16 + i]; },  // B^-1[i][t]
                                   [&](int t, int j) { return Ts[(16 * wv + j) * 33 + t]; }, lane);      // T[t][j]
#pragma unroll
    for (int r = 0; r < 4; ++r) Ms[(b0 + m) * LP + b0 + 16 + q + 4 * r] = -x[r];
  }
  lds_barrier();
  // level 64: one 16x16 tile of the 32x32 products per wave
  {
    const bool on = NW == 4 || wv < 4;
    const int tr = 16 * (wv & 1), tc = 16 * ((wv >> 1) & 1);
    if (on) {
      const double4_t t1 = lds_mma<8>([&](int i, int t) { return As[t * LP + 32 + tr + i]; },     // C[i][t]
                                      [&](int t, int j) { return Ms[(tc + j) * LP + t]; }, lane);  // A^-1[t][j]
#pragma unroll
      for (int r = 0; r < 4; ++r) Ts[(tc + m) * 33 + tr + q + 4 * r] = t1[r];
    }
    lds_barrier();
    if (on) {
      const double4_t x = lds_mma<8>([&](int i, int t) { return Ms[(32 + t) * LP + 32 + tr + i]; },  // B^-1[i][t]
                                     [&](int t, int j) { return Ts[(tc + j) * 33 + t]; }, lane);      // T[t][j]
#pragma unroll
      for (int r = 0; r < 4; ++r) Ms[(tc + m) * LP + 32 + tr + q + 4 * r] = -x[r];  // lower-left quadrant: read by nobody above
    }
  }
  lds_barrier();
}

__device__ __forceinline__ void potf2_inv_lds(Potf2Lds& sh) {
  potf2_factor_lds(sh, [] {});
  potf2_invert_lds(sh);
}

// potf2 + inverse for the chain workgroup of the dataflow launch (8 waves): the arithmetic of potf2_factor_lds +
// potf2_invert_lds, but the inversion does not wait for the last pivot.  Column panel b of L is final as soon as its 16
// pivots are done, so while wave 0 holds the pivots of the next panel (the other waves would idle), other waves invert the
// 16x16 diagonal blocks and build the products that need nothing newer; during the LAST panel wave 1 follows wave 0 column
// by column (inv16_chase), so the inverse of the last diagonal block is complete ~one inversion step after the last pivot.
// Behind the pivots there are two rounds of one 16x16x16 product left (the lower half of the big off-diagonal block is
// taken as -M33 (T_low - T' T_up) instead of -(X32 T_up + M33 T_low): it does not wait for X32):  ~0.6 us instead of ~3.9.
// `T2`: 16 x 17 doubles of scratch.  `side()` is run by the waves 1..7 during the last 16 pivots, behind their own work
// (they may watch sh.pivots_done), and by wave 0 behind its last pivot; `after_pivots()` by all waves behind the barrier
// that ends the pivots.  tid 0 must have cleared
// sh.pivots_done / next_ready / progress / x10_done (a barrier is passed before they are used).
template <typename FS, typename F>
__device__ __forceinline__ void potf2_chain_lds(Potf2Lds& sh, double* T2, int tid, FS&& side, F&& after_pivots,
                                                unsigned whatif = 0u) {
  double* As = sh.As;
  double* Ms = sh.Ms;
  double* Ts = sh.Ts;
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, q = lane >> 4;
  auto panel = [&](int k) {  // wave 0, lane = row
    double s[NBS], ri[NBS];
#pragma unroll
    for (int c = 0; c < NBS; ++c) s[c] = As[(k + c) * LP + lane];
    bool bad = false;
    panel16_factor<0>(s, ri, As, k, lane, bad);
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < NBS; ++c) sh.rinv[k + c] = ri[c];
    }
    if (bad) sh.bad = 1;
  };
  auto trailing = [&](int k) {  // rank-16 update of the trailing lower triangle: 16x16 tiles (ti >= tj) over the waves
    const int nb = (NBI - (k + NBS)) / NBS;
    for (int t = wv; t < nb * (nb + 1) / 2; t += 8) {
      const int ti = t < 1 ? 0 : (t < 3 ? 1 : 2), tj = t - ti * (ti + 1) / 2;
      const int rb = k + NBS + 16 * ti, cb = k + NBS + 16 * tj;
      const double4_t u = lds_mma<4>([&](int i, int tt) { return As[(k + tt) * LP + rb + i]; },
                                     [&](int tt, int j) { return As[(k + tt) * LP + cb + j]; }, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) As[(cb + m) * LP + rb + q + 4 * r] -= u[r];
    }
  };
  auto inv_diag = [&](int b0) {  // one wave: column m of M per lane (all four 16-lane rows redundantly)
    double acc[NBS];
#pragma unroll
    for (int j = 0; j < NBS; ++j) acc[j] = 0.0;
    inv16_lean<0>(acc, sh.rinv + b0, As + b0 * LP + b0, Ms + (b0 + m) * LP + b0, m, lane < NBS);
  };
  // inv([A 0; C B]) = [A^-1 0; -B^-1 C A^-1  B^-1] for the 32x32 block at b0: T = C A^-1 (into `t`, pitch tp), then -B^-1 T
  auto pair_t = [&](int b0, double* t, int tp) {
    const double4_t t1 = lds_mma<4>([&](int i, int tt) { return As[(b0 + tt) * LP + b0 + 16 + i]; },   // C[i][t]
                                    [&](int tt, int j) { return Ms[(b0 + j) * LP + b0 + tt]; }, lane);  // A^-1[t][j]
#pragma unroll
    for (int r = 0; r < 4; ++r) t[m * tp + q + 4 * r] = t1[r];
  };
  auto pair_x = [&](int b0, const double* t, int tp) {
    const double4_t x = lds_mma<4>([&](int i, int tt) { return Ms[(b0 + 16 + tt) * LP + b0 + 16 + i]; },  // B^-1[i][t]
                                   [&](int tt, int j) { return t[j * tp + tt]; }, lane);                   // T[t][j]
#pragma unroll
    for (int r = 0; r < 4; ++r) Ms[(b0 + m) * LP + b0 + 16 + q + 4 * r] = -x[r];
  };
  // 64 level, T = C A^-1 with C = L[32.., 0..31], A^-1 = the first 32x32 inverse: 16x16 tile (tr, tc) into Ts
  auto t64 = [&](int tr, int tc) {
    const double4_t t1 = lds_mma<8>([&](int i, int t) { return As[t * LP + 32 + tr + i]; },     // C[i][t]
                                    [&](int t, int j) { return Ms[(tc + j) * LP + t]; }, lane);  // A^-1[t][j]
#pragma unroll
    for (int r = 0; r < 4; ++r) Ts[(tc + m) * 33 + tr + q + 4 * r] = t1[r];
  };
  CHAIN_STAMP(20);
  if (wv == 0) panel(0);
  lds_barrier();
  CHAIN_STAMP(21);
  if (!FLOW_SKIP(whatif, 128u)) trailing(0);
  lds_barrier();
  CHAIN_STAMP(22);
  if (wv == 0) panel(16);
  else if (wv == 1 && !FLOW_SKIP(whatif, 64u)) inv_diag(0);
  lds_barrier();
  CHAIN_STAMP(23);
  if (!FLOW_SKIP(whatif, 128u)) trailing(16);
  lds_barrier();
  CHAIN_STAMP(24);
  if (wv == 0) panel(32);
  else if (wv == 1 && !FLOW_SKIP(whatif, 64u)) inv_diag(16);
  else if (wv == 2 && !FLOW_SKIP(whatif, 64u)) pair_t(0, Ts, 33);
  lds_barrier();
  CHAIN_STAMP(25);
  if (!FLOW_SKIP(whatif, 128u)) trailing(32);
  lds_barrier();
  CHAIN_STAMP(26);
  if (wv == 0) {
    double s[NBS], ri[NBS];
#pragma unroll
    for (int c = 0; c < NBS; ++c) s[c] = As[(48 + c) * LP + lane];
    bool bad = false;
    panel16_factor<0, true>(s, ri, As, 48, lane, bad, &sh.progress);
    if (bad) sh.bad = 1;
    if (lane == 0) *(volatile int*)&sh.pivots_done = 1;
    CHAIN_SINCE(56, 26);
    side();  // (does not wait any more: wave 0 asks for its part before it joins the others at the barrier)
  } else {
    if (FLOW_SKIP(whatif, 1u) && wv == 1) {
    } else if (FLOW_SKIP(whatif, 256u) && (wv == 2 || wv == 3 || wv == 6 || wv == 7)) {
    } else if (wv == 1) {  // the last diagonal block, column by column behind wave 0
      double acc[NBS];
#pragma unroll
      for (int j = 0; j < NBS; ++j) acc[j] = 0.0;
      inv16_chase<0>(acc, As + 48 * LP + 48, As + 48 * LP, &sh.progress, Ms + (48 + m) * LP + 48, m, lane < NBS);
      CHAIN_SINCE(57, 26);
    } else if (wv == 3) {
      inv_diag(32);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // M22 is in LDS before this wave reads it back
      pair_t(32, T2, 17);
      CHAIN_SINCE(58, 26);
    } else if (wv == 2 || wv == 6 || wv == 7) {  // (not wave 4: it shares its SIMD with wave 0)
      if (wv == 2) {
        pair_x(0, Ts, 33);
        asm volatile("" ::: "memory");
        if (lane == 0) *(volatile int*)&sh.x10_done = 1;
      }
      while (__hip_atomic_load(&sh.x10_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
      if (wv == 2) {
        t64(0, 0);
        t64(16, 0);
      } else {
        t64(wv == 6 ? 0 : 16, 16);
      }
      if (wv == 2) CHAIN_SINCE(59, 26);
      if (wv == 7) CHAIN_SINCE(60, 26);
    }
    side();
    if (wv == 5) CHAIN_SINCE(61, 26);
    if (wv == 6) CHAIN_SINCE(62, 26);
  }
  lds_barrier();
  CHAIN_STAMP(27);
  after_pivots();
  CHAIN_STAMP(28);
  if (FLOW_SKIP(whatif, 4u)) return;
  if (wv == 1) pair_x(32, T2, 17);
  else if (wv == 2 || wv == 3) {  // upper half of the big off-diagonal block: -M22 T_up
    const int tc = wv == 2 ? 0 : 16;
    const double4_t x = lds_mma<4>([&](int i, int t) { return Ms[(32 + t) * LP + 32 + i]; },
                                   [&](int t, int j) { return Ts[(tc + j) * 33 + t]; }, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) Ms[(tc + m) * LP + 32 + q + 4 * r] = -x[r];
  } else if (wv == 6 || wv == 7) {  // W = T_low - T' T_up, in place
    const int tc = wv == 6 ? 0 : 16;
    const double4_t w = lds_mma<4>([&](int i, int t) { return T2[t * 17 + i]; },
                                   [&](int t, int j) { return Ts[(tc + j) * 33 + t]; }, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) Ts[(tc + m) * 33 + 16 + q + 4 * r] -= w[r];
  }
  lds_barrier();
  CHAIN_STAMP(29);
  if (wv == 6 || wv == 7) {  // lower half: -M33 W
    const int tc = wv == 6 ? 0 : 16;
    const double4_t x = lds_mma<4>([&](int i, int t) { return Ms[(48 + t) * LP + 48 + i]; },
                                   [&](int t, int j) { return Ts[(tc + j) * 33 + 16 + t]; }, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) Ms[(tc + m) * LP + 48 + q + 4 * r] = -x[r];
  }
  lds_barrier();
  CHAIN_STAMP(30);
}
