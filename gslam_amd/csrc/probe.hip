// VALU issue-rate probes (bench.py's measured ceiling for the VALU-bound kernels).  One kernel per instruction class of
// the ORB / matcher inner loops: register-only chains, 8 independent accumulators per lane so that dependent-issue
// latency never limits, every SIMD holding 8 waves.  Result = wave-instructions per second over the whole device; the
// caller divides by (CUs x 4 SIMDs x clock) to see whether an instruction issues every 4 cycles (one pass of a wave64
// over the SIMD's 16 lanes), every 8 (two passes), or slower (quarter-rate multiplies).
#include "common.h"

namespace {

enum : int {
  kOpXorBcnt = 0,   // v_xor_b32 + v_bcnt_u32_b32 (the matcher's inner loop)
  kOpAdd,           // v_add_u32
  kOpPerm,          // v_perm_b32
  kOpPkMinMax,      // v_pk_max_i16 / v_pk_min_i16
  kOpMinMax3,       // v_max3_u32 / v_min3_u32
  kOpDot4,          // v_dot4_u32_u8
  kOpAlignbyte,     // v_alignbyte_b32
  kOpPkAdd,         // v_pk_add_u16
  kOpMad24,         // v_mad_u32_u24
  kOpLshlOr,        // v_lshl_or_b32
  kOpMulLo,         // v_mul_lo_u32 (quarter-rate reference point)
  kOpFastMix,       // perm, pk_max, pk_min, max3, min3, dot4, alignbyte, add interleaved (orb_fast_cells' mix)
  // which property sets the rate: the datapath (fp32 vs integer) or the encoding (32-bit VOP2 vs 64-bit VOP3 / VOP3P)?
  kOpXor,           // v_xor_b32        VOP2, integer
  kOpMaxU32,        // v_max_u32        VOP2, integer
  kOpMul24,         // v_mul_u32_u24    VOP2, integer multiplier
  kOpAddF32,        // v_add_f32        VOP2, fp32
  kOpFmaF32,        // v_fma_f32        VOP3, fp32 (the guide's 2-clock instruction)
  kOpPkFmaF32,      // v_pk_fma_f32     VOP3P, two fp32 per lane
  kOpAdd3,          // v_add3_u32       VOP3, integer
  kOpBcnt,          // v_bcnt_u32_b32   VOP3, integer
  kOpCount
};

const char* const kOpNames[kOpCount] = {"xor+bcnt", "add_u32", "perm_b32", "pk_max/min_i16", "max3/min3_u32", "dot4_u32_u8",
                                        "alignbyte_b32", "pk_add_u16", "mad_u32_u24", "lshl_or_b32", "mul_lo_u32",
                                        "fast_cells mix", "xor_b32 (VOP2)", "max_u32 (VOP2)", "mul_u32_u24 (VOP2)",
                                        "add_f32 (VOP2)", "fma_f32 (VOP3)", "pk_fma_f32 (VOP3P)", "add3_u32 (VOP3)",
                                        "bcnt_u32_b32 (VOP3)"};

template <int OP>
__global__ __launch_bounds__(256) void valu_issue_kernel(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a[8], b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x01020304u;
  uint64_t a64[4] = {seed, seed + 1, seed + 2, seed + 3};  // register pairs of the packed-fp32 probe
  const uint64_t b64 = ((uint64_t)b << 32) | c, c64 = ((uint64_t)c << 32) | b;
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = seed * (k + 1) + threadIdx.x * 97u;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (OP == kOpXorBcnt) {
          if (u & 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
          else asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
        }
        if (OP == kOpAdd) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
        if (OP == kOpPerm) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (OP == kOpPkMinMax) {
          if (u & 1) asm volatile("v_pk_min_i16 %0, %1, %0" : "+v"(a[k]) : "v"(b));
          else asm volatile("v_pk_max_i16 %0, %1, %0" : "+v"(a[k]) : "v"(b));
        }
        if (OP == kOpMinMax3) {
          if (u & 1) asm volatile("v_min3_u32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
          else asm volatile("v_max3_u32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
        }
        if (OP == kOpDot4) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
        if (OP == kOpAlignbyte) asm volatile("v_alignbyte_b32 %0, %1, %0, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (OP == kOpPkAdd) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(a[k]) : "v"(b));
        if (OP == kOpMad24) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
        if (OP == kOpLshlOr) asm volatile("v_lshl_or_b32 %0, %1, 3, %0" : "+v"(a[k]) : "v"(b));
        if (OP == kOpMulLo) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
        if (OP == kOpXor) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
        if (OP == kOpMaxU32) asm volatile("v_max_u32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
        if (OP == kOpMul24) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(a[k]) : "v"(b));
        if (OP == kOpAddF32) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
        if (OP == kOpFmaF32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
        if (OP == kOpAdd3) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
        if (OP == kOpBcnt) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
        if (OP == kOpPkFmaF32 && k < 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a64[k]) : "v"(b64), "v"(c64));
        if (OP == kOpFastMix) {
          switch ((u * 8 + k) & 7) {
            case 0: asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(a[k]) : "v"(b), "v"(c)); break;
            case 1: asm volatile("v_pk_max_i16 %0, %1, %0" : "+v"(a[k]) : "v"(b)); break;
            case 2: asm volatile("v_pk_min_i16 %0, %1, %0" : "+v"(a[k]) : "v"(b)); break;
            case 3: asm volatile("v_max3_u32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c)); break;
            case 4: asm volatile("v_min3_u32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c)); break;
            case 5: asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c)); break;
            case 6: asm volatile("v_alignbyte_b32 %0, %1, %0, %2" : "+v"(a[k]) : "v"(b), "v"(c)); break;
            default: asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[k]) : "v"(b)); break;
          }
        }
      }
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += a[k];
  for (int k = 0; k < 4; ++k) s += (uint32_t)a64[k] + (uint32_t)(a64[k] >> 32);
  if (s == 0xFFFFFFFFu) out[0] = s;
}

template <int OP>
gh_status run_probe(gh_ctx* ctx, uint32_t* out, double* rate) {
  const int iters = 2048, blocks = ctx->cu_count > 0 ? ctx->cu_count * 8 : 2048;  // 8 blocks x 4 waves = 8 waves / SIMD
  hipEvent_t e0, e1;
  GH_HIP(ctx, hipEventCreate(&e0));
  GH_HIP(ctx, hipEventCreate(&e1));
  hipLaunchKernelGGL(valu_issue_kernel<OP>, dim3(blocks), dim3(256), 0, ctx->stream, out, 16, 12345u);  // warm-up
  GH_HIP(ctx, hipEventRecord(e0, ctx->stream));
  hipLaunchKernelGGL(valu_issue_kernel<OP>, dim3(blocks), dim3(256), 0, ctx->stream, out, iters, 12345u);
  GH_HIP(ctx, hipEventRecord(e1, ctx->stream));
  GH_HIP(ctx, hipEventSynchronize(e1));
  float ms = 0.f;
  GH_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *rate = (double)blocks * 4.0 * (double)iters * 32.0 / (ms * 1e-3);  // wave-instructions per second
  return GH_OK;
}

}  // namespace

extern "C" gh_status gh_valu_issue_probe(gh_ctx* ctx, int op, double* wave_insts_per_s, char* name, int name_cap) {
  if (!ctx || !wave_insts_per_s) return GH_ERR_ARG;
  GH_ENTER(ctx);
  if (op < 0 || op >= kOpCount) return GH_ERR_ARG;  // callers iterate op = 0, 1, ... until this
  if (name && name_cap > 0) snprintf(name, name_cap, "%s", kOpNames[op]);
  void* out = nullptr;
  GH_TRY(gh_scratch(ctx, 256, &out));
  uint32_t* o = (uint32_t*)out;
  switch (op) {
    case kOpXorBcnt: return run_probe<kOpXorBcnt>(ctx, o, wave_insts_per_s);
    case kOpAdd: return run_probe<kOpAdd>(ctx, o, wave_insts_per_s);
    case kOpPerm: return run_probe<kOpPerm>(ctx, o, wave_insts_per_s);
    case kOpPkMinMax: return run_probe<kOpPkMinMax>(ctx, o, wave_insts_per_s);
    case kOpMinMax3: return run_probe<kOpMinMax3>(ctx, o, wave_insts_per_s);
    case kOpDot4: return run_probe<kOpDot4>(ctx, o, wave_insts_per_s);
    case kOpAlignbyte: return run_probe<kOpAlignbyte>(ctx, o, wave_insts_per_s);
    case kOpPkAdd: return run_probe<kOpPkAdd>(ctx, o, wave_insts_per_s);
    case kOpMad24: return run_probe<kOpMad24>(ctx, o, wave_insts_per_s);
    case kOpLshlOr: return run_probe<kOpLshlOr>(ctx, o, wave_insts_per_s);
    case kOpMulLo: return run_probe<kOpMulLo>(ctx, o, wave_insts_per_s);
    case kOpXor: return run_probe<kOpXor>(ctx, o, wave_insts_per_s);
    case kOpMaxU32: return run_probe<kOpMaxU32>(ctx, o, wave_insts_per_s);
    case kOpMul24: return run_probe<kOpMul24>(ctx, o, wave_insts_per_s);
    case kOpAddF32: return run_probe<kOpAddF32>(ctx, o, wave_insts_per_s);
    case kOpFmaF32: return run_probe<kOpFmaF32>(ctx, o, wave_insts_per_s);
    case kOpPkFmaF32: {
      const gh_status st = run_probe<kOpPkFmaF32>(ctx, o, wave_insts_per_s);
      *wave_insts_per_s *= 0.5;  // the chain issues one v_pk_fma_f32 per TWO accumulators
      return st;
    }
    case kOpAdd3: return run_probe<kOpAdd3>(ctx, o, wave_insts_per_s);
    case kOpBcnt: return run_probe<kOpBcnt>(ctx, o, wave_insts_per_s);
    case kOpFastMix: return run_probe<kOpFastMix>(ctx, o, wave_insts_per_s);
    default: return GH_ERR_ARG;
  }
}
