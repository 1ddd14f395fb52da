// Issue-rate survey of gfx950 VALU instruction classes (perf tool, not part of libgslam_hip.so): which instructions run
// on the full-rate datapath (~2.6 clocks per wave64 instruction at the reported 2.4 GHz) and which at half rate (~4.4)?
// Register-only chains, 8 independent accumulators per lane, 8 waves per SIMD, every CU.
//   hipcc --offload-arch=gfx950 -O3 -o build/issue_probe tools/issue_probe.hip && build/issue_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define PROBE(NAME, ASM2)                                                                                      \
  __global__ __launch_bounds__(256) void k_##NAME(uint32_t* out, int iters, uint32_t seed) {                  \
    uint32_t a[8], b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x01020304u;                               \
    for (int k = 0; k < 8; ++k) a[k] = seed * (k + 1) + threadIdx.x * 97u;                                     \
    for (int i = 0; i < iters; ++i) {                                                                          \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                          \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile(ASM2 : "+v"(a[k]) : "v"(b), "v"(c));        \
      }                                                                                                        \
    }                                                                                                          \
    uint32_t s = 0;                                                                                            \
    for (int k = 0; k < 8; ++k) s += a[k];                                                                     \
    if (s == 0xFFFFFFFFu) out[0] = s;                                                                          \
  }

PROBE(xor_b32, "v_xor_b32 %0, %1, %0")
PROBE(and_b32, "v_and_b32 %0, %1, %0")
PROBE(or_b32, "v_or_b32 %0, %1, %0")
PROBE(add_u32, "v_add_u32 %0, %1, %0")
PROBE(sub_u32, "v_sub_u32 %0, %1, %0")
PROBE(lshlrev_b32, "v_lshlrev_b32 %0, 3, %0")
PROBE(lshrrev_b32, "v_lshrrev_b32 %0, 3, %0")
PROBE(ashrrev_i32, "v_ashrrev_i32 %0, 3, %0")
PROBE(mov_b32, "v_mov_b32 %0, %1")
PROBE(min_u32, "v_min_u32 %0, %1, %0")
PROBE(max_i32, "v_max_i32 %0, %1, %0")
PROBE(min_f32, "v_min_f32 %0, %1, %0")
PROBE(max_f32, "v_max_f32 %0, %1, %0")
PROBE(min3_f32, "v_min3_f32 %0, %1, %2, %0")
PROBE(max3_f32, "v_max3_f32 %0, %1, %2, %0")
PROBE(med3_f32, "v_med3_f32 %0, %1, %2, %0")
PROBE(min3_u32, "v_min3_u32 %0, %1, %2, %0")
PROBE(add_f32, "v_add_f32 %0, %1, %0")
PROBE(sub_f32, "v_sub_f32 %0, %1, %0")
PROBE(mul_f32, "v_mul_f32 %0, %1, %0")
PROBE(fma_f32, "v_fma_f32 %0, %1, %2, %0")
PROBE(mac_f32, "v_fmac_f32 %0, %1, %2")
PROBE(cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
PROBE(cvt_f32_ubyte3, "v_cvt_f32_ubyte3 %0, %0")
PROBE(cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
PROBE(cvt_u32_f32, "v_cvt_u32_f32 %0, %0")
PROBE(cndmask, "v_cndmask_b32 %0, %1, %0, vcc")
PROBE(cmp_gt_f32, "v_cmp_gt_f32 vcc, %1, %0\n v_xor_b32 %0, %2, %0")
PROBE(cmp_gt_u32, "v_cmp_gt_u32 vcc, %1, %0\n v_xor_b32 %0, %2, %0")
PROBE(bfe_u32, "v_bfe_u32 %0, %0, 3, 9")
PROBE(and_or_b32, "v_and_or_b32 %0, %1, %2, %0")
PROBE(or3_b32, "v_or3_b32 %0, %1, %2, %0")
PROBE(xad_u32, "v_xad_u32 %0, %1, %2, %0")
PROBE(add3_u32, "v_add3_u32 %0, %1, %2, %0")
PROBE(lshl_add_u32, "v_lshl_add_u32 %0, %1, 2, %0")
PROBE(add_lshl_u32, "v_add_lshl_u32 %0, %1, %0, 1")
PROBE(lshl_or_b32, "v_lshl_or_b32 %0, %1, 3, %0")
PROBE(bcnt, "v_bcnt_u32_b32 %0, %1, %0")
PROBE(perm_b32, "v_perm_b32 %0, %1, %0, %2")
PROBE(alignbit, "v_alignbit_b32 %0, %1, %0, 7")
PROBE(alignbyte, "v_alignbyte_b32 %0, %1, %0, 1")
PROBE(sad_u8, "v_sad_u8 %0, %1, %2, %0")
PROBE(msad_u8, "v_msad_u8 %0, %1, %2, %0")
PROBE(sad_u16, "v_sad_u16 %0, %1, %2, %0")
PROBE(dot4_u32_u8, "v_dot4_u32_u8 %0, %1, %2, %0")
PROBE(dot2_u32_u16, "v_dot2_u32_u16 %0, %1, %2, %0")
PROBE(mul_u32_u24, "v_mul_u32_u24 %0, %1, %0")
PROBE(mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0")
PROBE(mul_lo_u32, "v_mul_lo_u32 %0, %1, %0")
PROBE(max_u16, "v_max_u16 %0, %1, %0")
PROBE(sub_u16, "v_sub_u16 %0, %1, %0")
PROBE(pk_max_i16, "v_pk_max_i16 %0, %1, %0")
PROBE(pk_add_u16, "v_pk_add_u16 %0, %1, %0")
PROBE(pk_sub_i16, "v_pk_sub_i16 %0, %1, %0")
PROBE(pk_add_f16, "v_pk_add_f16 %0, %1, %0")
PROBE(pk_max_f16, "v_pk_max_f16 %0, %1, %0")
PROBE(pk_fma_f16, "v_pk_fma_f16 %0, %1, %2, %0")
PROBE(max_f16, "v_max_f16 %0, %1, %0")
PROBE(sdwa_max_u16_b0, "v_max_u16_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD")
PROBE(sdwa_add_u32_b1, "v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
PROBE(sdwa_max_f32, "v_max_f32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD")
PROBE(dpp_add_u32, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
PROBE(readlane_pair, "v_readlane_b32 s20, %1, 3\n v_xor_b32 %0, s20, %0")
PROBE(mix_xor_bcnt, "v_xor_b32 %0, %1, %0\n v_bcnt_u32_b32 %0, %2, %0")
PROBE(mix_fma_perm, "v_fma_f32 %0, %1, %2, %0\n v_perm_b32 %0, %1, %0, %2")
PROBE(mix_xor_xor_bcnt, "v_xor_b32 %0, %1, %0\n v_xor_b32 %0, %2, %0\n v_bcnt_u32_b32 %0, %2, %0")

// gfx950 additions and the classes the packed fp16 arc score of orb_fast_cells uses (round 3, end)
PROBE(pk_minimum3_f16, "v_pk_minimum3_f16 %0, %1, %2, %0")
PROBE(pk_maximum3_f16, "v_pk_maximum3_f16 %0, %1, %2, %0")
PROBE(minimum3_f32, "v_minimum3_f32 %0, %1, %2, %0")
PROBE(pk_min_f16, "v_pk_min_f16 %0, %1, %0")
PROBE(pk_min_u16, "v_pk_min_u16 %0, %1, %0")
PROBE(pk_mad_u16, "v_pk_mad_u16 %0, %1, %2, %0")
PROBE(mad_i32_i24, "v_mad_i32_i24 %0, %1, %2, %0")
PROBE(bitop3_b32, "v_bitop3_b32 %0, %1, %2, %0 bitop3:0x96")
PROBE(mix_pkmin3_perm, "v_pk_minimum3_f16 %0, %1, %2, %0\n v_perm_b32 %0, %1, %0, %2")
PROBE(mix_pkmin3_pkmax_i16, "v_pk_minimum3_f16 %0, %1, %2, %0\n v_pk_max_i16 %0, %1, %0")
PROBE(mix_pkmin3_fma_f32, "v_pk_minimum3_f16 %0, %1, %2, %0\n v_fma_f32 %0, %1, %2, %0")
PROBE(mix_pkmin3_xor, "v_pk_minimum3_f16 %0, %1, %2, %0\n v_xor_b32 %0, %1, %0")

struct P { const char* name; void (*fn)(uint32_t*, int, uint32_t); int per; };
#define E(NAME, PER) {#NAME, k_##NAME, PER}

int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount * 8, iters = 2048;
  uint32_t* out;
  CK(hipMalloc(&out, 256));
  const P probes[] = {E(xor_b32, 1), E(and_b32, 1), E(or_b32, 1), E(add_u32, 1), E(sub_u32, 1), E(lshlrev_b32, 1), E(lshrrev_b32, 1),
                      E(ashrrev_i32, 1), E(mov_b32, 1), E(min_u32, 1), E(max_i32, 1), E(min_f32, 1), E(max_f32, 1), E(min3_f32, 1),
                      E(max3_f32, 1), E(med3_f32, 1), E(min3_u32, 1), E(add_f32, 1), E(sub_f32, 1), E(mul_f32, 1), E(fma_f32, 1),
                      E(mac_f32, 1), E(cvt_f32_ubyte0, 1), E(cvt_f32_ubyte3, 1), E(cvt_f32_u32, 1), E(cvt_u32_f32, 1), E(cndmask, 1),
                      E(cmp_gt_f32, 2), E(cmp_gt_u32, 2), E(bfe_u32, 1), E(and_or_b32, 1), E(or3_b32, 1), E(xad_u32, 1), E(add3_u32, 1),
                      E(lshl_add_u32, 1), E(add_lshl_u32, 1), E(lshl_or_b32, 1), E(bcnt, 1), E(perm_b32, 1), E(alignbit, 1),
                      E(alignbyte, 1), E(sad_u8, 1), E(msad_u8, 1), E(sad_u16, 1), E(dot4_u32_u8, 1), E(dot2_u32_u16, 1),
                      E(mul_u32_u24, 1), E(mad_u32_u24, 1), E(mul_lo_u32, 1), E(max_u16, 1), E(sub_u16, 1), E(pk_max_i16, 1),
                      E(pk_add_u16, 1), E(pk_sub_i16, 1), E(pk_add_f16, 1), E(pk_max_f16, 1), E(pk_fma_f16, 1), E(max_f16, 1),
                      E(sdwa_max_u16_b0, 1), E(sdwa_add_u32_b1, 1), E(sdwa_max_f32, 1), E(dpp_add_u32, 1), E(readlane_pair, 2),
                      E(mix_xor_bcnt, 2), E(mix_fma_perm, 2), E(mix_xor_xor_bcnt, 3), E(pk_minimum3_f16, 1), E(pk_maximum3_f16, 1),
                      E(minimum3_f32, 1), E(pk_min_f16, 1), E(pk_min_u16, 1), E(pk_mad_u16, 1), E(mad_i32_i24, 1), E(bitop3_b32, 1),
                      E(mix_pkmin3_perm, 2), E(mix_pkmin3_pkmax_i16, 2), E(mix_pkmin3_fma_f32, 2), E(mix_pkmin3_xor, 2)};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double simd_clk = (double)prop.multiProcessorCount * 4.0 * prop.clockRate * 1e3;
  printf("%s: %d CUs, reported clock %.0f MHz; clocks per wave64 instruction assume that clock\n", prop.gcnArchName,
         prop.multiProcessorCount, prop.clockRate / 1e3);
  for (const P& p : probes) {
    hipLaunchKernelGGL(p.fn, dim3(blocks), dim3(256), 0, 0, out, 16, 12345u);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(p.fn, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double rate = (double)blocks * 4.0 * iters * 32.0 * p.per / (ms * 1e-3);
    printf("%-20s %7.1f G wave-inst/s  %5.2f clk%s\n", p.name, rate / 1e9, simd_clk / rate, p.per > 1 ? "  (average over the group)" : "");
  }
  return 0;
}
