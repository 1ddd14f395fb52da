// f64 global atomicAdd throughput on gfx950 by address pattern (perf tool, not part of libgslam_hip.so): what bounds the
// atomics of the general graph solver (posegraph.hip: gr_obs_lin / gr_schur add 7 x 7 blocks into a dense H)?
//   scatter   every lane of an instruction hits its own 64-byte sector of a 6 MB array (what the solver does today)
//   run7      groups of 8 lanes hit 7 consecutive doubles of one block column (56 B), groups scattered
//   run8x8    a wave hits 8 full 64-byte sectors
//   same      all lanes, all waves: one word
//   lds_then  a workgroup first sums 64 words in LDS, then issues one global atomic per word
// Each thread issues `iters` atomics; rate = lane-atomics per second.
//   hipcc --offload-arch=gfx950 -O3 -o build/atomic_probe tools/atomic_probe.hip && build/atomic_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kWords = 840 * 848;  // the reduced system of 120 keyframes

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_atomic(double* H, int iters) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  __shared__ double acc[64];
  if (MODE == 4 && threadIdx.x < 64) acc[threadIdx.x] = 0.0;
  if (MODE == 4) __syncthreads();
  for (int i = 0; i < iters; ++i) {
    uint32_t idx;
    if (MODE == 0) idx = (mix(t * 131u + i) % (kWords / 8)) * 8 + (t & 7);                     // own sector per lane
    else if (MODE == 1) idx = (mix((t >> 3) * 131u + i) % (kWords / 8)) * 8 + (t & 7);         // 8 lanes -> one sector (lane 7 idle below)
    else if (MODE == 2) idx = (mix((t >> 3) * 131u + i) % (kWords / 8)) * 8 + (t & 7);         // same, all 8 lanes
    else if (MODE == 3) idx = 0;
    else idx = threadIdx.x & 63;
    if (MODE == 1 && (t & 7) == 7) continue;
    if (MODE == 4) atomicAdd(&acc[idx], 1.0);
    else atomicAdd(&H[idx], 1.0);
  }
  if (MODE == 4) {
    __syncthreads();
    if (threadIdx.x < 64) atomicAdd(&H[(mix(blockIdx.x) % (kWords / 64)) * 64 + threadIdx.x], acc[threadIdx.x]);
  }
}

template <int MODE>
static void run(const char* name, double* H, int blocks, int iters, double lanes_frac) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k_atomic<MODE>, dim3(blocks), dim3(256), 0, 0, H, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k_atomic<MODE>, dim3(blocks), dim3(256), 0, 0, H, iters);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double n = (double)blocks * 256 * iters * lanes_frac;
  printf("%-10s %8.3f ms  %7.2f G lane-atomics/s\n", name, ms, n / ms * 1e-6);
}

int main() {
  double* H;
  CK(hipMalloc(&H, (size_t)kWords * 8));
  CK(hipMemset(H, 0, (size_t)kWords * 8));
  const int blocks = 2048, iters = 64;
  run<0>("scatter", H, blocks, iters, 1.0);
  run<1>("run7", H, blocks, iters, 7.0 / 8.0);
  run<2>("run8x8", H, blocks, iters, 1.0);
  run<3>("same", H, blocks, 8, 1.0);
  run<4>("lds_then", H, blocks, iters, 1.0);
  return 0;
}
