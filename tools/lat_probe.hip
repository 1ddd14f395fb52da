// Dependent-issue latency of the instructions on the pivot chain of potf2 (one wave, one SIMD; perf probe only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/lat_probe.hip -o build/lat_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ long long g_out[16];
__device__ double g_sink;

template <int OP>
__global__ void lat_kernel(double seed, int iters) {
  double x = seed + threadIdx.x * 1e-3, y = 1.0000001;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (OP == 0) x = __builtin_fma(x, y, 1e-9);                 // v_fma_f64 chain
      if (OP == 1) x = x * y;                                     // v_mul_f64 chain
      if (OP == 2) x = __builtin_amdgcn_rsq(x) + 1.5;             // v_rsq_f64 + v_add_f64
      if (OP == 3) {                                              // readlane pair -> fma with SGPR operand
        const int lo = __builtin_amdgcn_readlane(__double2loint(x), 5), hi = __builtin_amdgcn_readlane(__double2hiint(x), 5);
        x = __builtin_fma(x, 1e-9, __hiloint2double(hi, lo));
      }
      if (OP == 4) x = __builtin_amdgcn_rsq(x);                   // v_rsq_f64 chain alone
      if (OP == 5) {                                              // float fma chain for comparison
        float f = (float)x;
#pragma unroll
        for (int v = 0; v < 4; ++v) f = __builtin_fmaf(f, 1.0000001f, 1e-9f);
        x = (double)f;
      }
      if (OP == 6) x = (threadIdx.x > 3) ? x * y : 0.0;           // mul + 2 cndmask
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) g_out[OP] = t1 - t0;
  if (x == 12345.678) g_sink = x;
}

// 8 independent chains interleaved: issue rate of one wave (cycles per instruction)
template <int OP>
__global__ void ilp_kernel(double seed, int iters) {
  double x[8];
  for (int c = 0; c < 8; ++c) x[c] = seed + threadIdx.x * 1e-3 + c;
  const double y = 1.0000001;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (OP == 0) x[c] = __builtin_fma(x[c], y, 1e-9);
        if (OP == 1) {
          const int lo = __builtin_amdgcn_readlane(__double2loint(x[c]), 5), hi = __builtin_amdgcn_readlane(__double2hiint(x[c]), 5);
          x[c] = __builtin_fma(x[c], 1e-9, __hiloint2double(hi, lo));
        }
        if (OP == 2) x[c] = __builtin_amdgcn_rsq(x[c]);
      }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) g_out[8 + OP] = t1 - t0;
  double t = 0;
  for (int c = 0; c < 8; ++c) t += x[c];
  if (t == 12345.678) g_sink = t;
}

int main() {
  const int iters = 256;
  const char* names[] = {"v_fma_f64", "v_mul_f64", "v_rsq_f64 + v_add_f64", "2 x v_readlane + v_fma_f64(sgpr)", "v_rsq_f64",
                         "cvt + 4 v_fma_f32 + cvt", "v_mul_f64 + 2 v_cndmask"};
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(lat_kernel<0>, dim3(1), dim3(64), 0, 0, 1.5, iters);
    hipLaunchKernelGGL(lat_kernel<1>, dim3(1), dim3(64), 0, 0, 1.5, iters);
    hipLaunchKernelGGL(lat_kernel<2>, dim3(1), dim3(64), 0, 0, 1.5, iters);
    hipLaunchKernelGGL(lat_kernel<3>, dim3(1), dim3(64), 0, 0, 1.5, iters);
    hipLaunchKernelGGL(lat_kernel<4>, dim3(1), dim3(64), 0, 0, 1.5, iters);
    hipLaunchKernelGGL(lat_kernel<5>, dim3(1), dim3(64), 0, 0, 1.5, iters);
    hipLaunchKernelGGL(lat_kernel<6>, dim3(1), dim3(64), 0, 0, 1.5, iters);
    hipLaunchKernelGGL(ilp_kernel<0>, dim3(1), dim3(64), 0, 0, 1.5, iters);
    hipLaunchKernelGGL(ilp_kernel<1>, dim3(1), dim3(64), 0, 0, 1.5, iters);
    hipLaunchKernelGGL(ilp_kernel<2>, dim3(1), dim3(64), 0, 0, 1.5, iters);
    hipDeviceSynchronize();
    long long o[16];
    hipMemcpyFromSymbol(o, HIP_SYMBOL(g_out), sizeof(o));
    for (int k = 0; k < 7; ++k) printf("rep %d  %-36s %.1f cycles per dependent step\n", rep, names[k], (double)o[k] / (iters * 16));
    printf("rep %d  8 independent v_fma_f64 chains: %.1f cycles per instruction\n", rep, (double)o[8] / (iters * 16));
    printf("rep %d  8 independent (2 readlane + fma): %.1f cycles per group of 3\n", rep, (double)o[9] / (iters * 16));
    printf("rep %d  8 independent v_rsq_f64: %.1f cycles per instruction\n", rep, (double)o[10] / (iters * 16));
  }
  return 0;
}
