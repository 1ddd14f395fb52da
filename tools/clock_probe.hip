// Sustained core clock under a dense f64 MFMA load (perf tool, not part of libgslam_hip.so): every CU runs
// v_mfma_f64_16x16x4_f64 back to back for ~2 s; workgroup 0 samples the shader clock (s_memtime) against the constant
// 100 MHz wall clock (s_memrealtime) every segment.  Answers: is a 60 000 x 60 000 Cholesky at 84 % of the nominal MFMA
// peak short of the hardware, or at the clock the part sustains under that load?
//   hipcc --offload-arch=gfx950 -O3 -o build/clock_probe tools/clock_probe.hip && build/clock_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void load_kernel(unsigned long long* samples, int segments, int iters, double seed, double* sink,
                                                   int heavy) {
  double4_t acc[8];
  for (int k = 0; k < 8; ++k) acc[k] = (double4_t){seed * k, 1.0, 2.0, 3.0};
  const double a = seed + threadIdx.x * 1e-9, b = 1.0 + threadIdx.x * 1e-9;
  for (int s = 0; s < segments; ++s) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      samples[2 * s] = clock64();
      samples[2 * s + 1] = wall_clock64();
    }
    for (int i = 0; i < iters; ++i) {
      if (heavy) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
      } else {
        __builtin_amdgcn_s_sleep(64);
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    samples[2 * segments] = clock64();
    samples[2 * segments + 1] = wall_clock64();
  }
  double t = 0;
  for (int k = 0; k < 8; ++k) t += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  if (t == 12345.678) sink[0] = t;
}

int main(int argc, char** argv) {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int segments = 40;
  unsigned long long* d;
  double* sink;
  CK(hipMalloc(&d, (2 * segments + 2) * 8));
  CK(hipMalloc(&sink, 8));
  unsigned long long h[2 * 40 + 2];
  printf("%s: %d CUs, reported clock %.0f MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1e3);
  for (int heavy = 0; heavy < 2; ++heavy) {
    // heavy: 8 waves per CU (2 per SIMD), 8 independent accumulators each: the matrix core never idles
    const int iters = heavy ? (argc > 1 ? atoi(argv[1]) : 200000) : 2000;
    hipLaunchKernelGGL(load_kernel, dim3(prop.multiProcessorCount * (argc > 2 ? atoi(argv[2]) : 2)), dim3(256), 0, 0, d, segments, iters, 1.0, sink, heavy);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    printf("%s load: shader clock per segment (MHz):", heavy ? "f64 MFMA on every CU" : "idle (s_sleep)");
    double total_ms = 0, flops = 0;
    for (int s = 0; s < segments; ++s) {
      const double dt = (double)(h[2 * s + 3] - h[2 * s + 1]) / 100e6;  // seconds
      const double mhz = (double)(h[2 * s + 2] - h[2 * s]) / dt / 1e6;
      total_ms += dt * 1e3;
      if (s % 4 == 0) printf(" %.0f", mhz);
      if (heavy && s % 4 == 0)
        printf("[%.1f TF]", (double)prop.multiProcessorCount * (argc > 2 ? atoi(argv[2]) : 2) * 4 * (double)iters * 8 * 2048.0 / dt / 1e12);
    }
    if (heavy) {
      flops = (double)prop.multiProcessorCount * (argc > 2 ? atoi(argv[2]) : 2) * 4 * (double)segments * iters * 8 * 2048.0;
      printf("\n   %.1f ms, %.1f TFLOP/s of f64 MFMA (nominal %.1f at the reported clock)\n", total_ms, flops / (total_ms * 1e-3) / 1e12,
             prop.multiProcessorCount * 128.0 * prop.clockRate * 1e3 / 1e12);
    } else {
      printf("\n   %.1f ms\n", total_ms);
    }
  }
  return 0;
}
