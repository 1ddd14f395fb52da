// Rate of v_mfma_i32_16x16x64_i8 on gfx950 (perf tool, not part of libgslam_hip.so): independent accumulators vs a
// dependent chain, by waves per SIMD, with and without VALU instructions issued in the shadow of the matrix core.
//   hipcc --offload-arch=gfx950 -O3 -o build/mfma_probe tools/mfma_probe.hip && build/mfma_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

// NACC independent accumulators, VALU_PER extra VALU instructions (v_min_u32 + v_med3_u32 pairs) after every MFMA
template <int NACC, int VALU_PER, int KIND = 0>
__global__ __launch_bounds__(256) void k_mfma(int* out, int iters, int seed) {
  v4i a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed * 3, seed * 5, seed * 7, seed * 9};
  v4i acc[NACC];
  uint32_t x[8];
  for (int k = 0; k < 8; ++k) x[k] = seed * (k + 1) + threadIdx.x;
  for (int k = 0; k < NACC; ++k) acc[k] = v4i{k, k, k, k};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u) {
#pragma unroll
      for (int k = 0; k < NACC; ++k) {
        asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
#pragma unroll
        for (int v = 0; v < VALU_PER; ++v)
          if (KIND == 0) asm volatile("v_min_u32 %0, %1, %0" : "+v"(x[(2 * k + v) & 7]) : "v"(seed));
          else if (KIND == 1) asm volatile("v_med3_u32 %0, %1, %2, %0" : "+v"(x[(2 * k + v) & 7]) : "v"(seed), "v"(x[(2 * k + v + 3) & 7]));
          else if (KIND == 2) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x[(2 * k + v) & 7]) : "v"(seed));
          else asm volatile("v_med3_u32 %0, %1, %2, %0" : "+v"(x[(2 * k + v) & 7]) : "v"(acc[(k + 2) % NACC][v & 3]), "v"(x[(2 * k + v + 3) & 7]));
      }
    }
  }
  int s = 0;
  for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  for (int k = 0; k < 8; ++k) s += x[k];
  if (s == 0x7FFFFFFF) out[0] = s;
}

typedef void (*fn_t)(int*, int, int);
struct P { const char* name; fn_t fn; };

int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  int* out;
  CK(hipMalloc(&out, 256));
  const P probes[] = {{"1 chain (dependent)", k_mfma<1, 0>}, {"2 chains", k_mfma<2, 0>}, {"4 chains", k_mfma<4, 0>}, {"8 chains", k_mfma<8, 0>},
                      {"4 chains + 2 VALU each", k_mfma<4, 2>}, {"4 chains + 3 VALU each", k_mfma<4, 3>},
                      {"4 chains + 4 VALU each", k_mfma<4, 4>}, {"1 chain + 3 VALU each", k_mfma<1, 3>},
                      {"4 chains + 2 med3 each", k_mfma<4, 2, 1>}, {"4 chains + 4 med3 each", k_mfma<4, 4, 1>},
                      {"4 chains + 4 and_b32 each", k_mfma<4, 4, 2>}, {"4 chains + 6 and_b32 each", k_mfma<4, 6, 2>},
                      {"4 chains + 2 med3(acc) each", k_mfma<4, 2, 3>}, {"4 chains + 4 med3(acc) each", k_mfma<4, 4, 3>}};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double simd_clk = (double)prop.multiProcessorCount * 4.0 * prop.clockRate * 1e3;
  printf("%s: %d CUs, reported clock %.0f MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1e3);
  const int iters = 4096;
  for (int wps : {2, 4}) {  // waves per SIMD
    for (const P& p : probes) {
      const int blocks = prop.multiProcessorCount * wps;
      hipLaunchKernelGGL(p.fn, dim3(blocks), dim3(256), 0, 0, out, 16, 3);
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(p.fn, dim3(blocks), dim3(256), 0, 0, out, iters, 3);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double mfma = (double)blocks * 4.0 * iters * 16.0;
      printf("%d waves/SIMD  %-26s %7.1f us  %6.2f clk per MFMA per SIMD  %7.1f TOPS\n", wps, p.name, ms * 1e3,
             simd_clk * (ms * 1e-3) / mfma, mfma * 32768.0 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
