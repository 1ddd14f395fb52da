// Host <-> HBM link probe for the host-fed extraction path (gh_orb_stream_*): what does THIS box's PCIe link deliver,
// and through which mechanism?  Perf tool only (not part of libgslam_hip.so).
//   hipcc --offload-arch=gfx950 -O3 -o build/pcie_probe tools/pcie_probe.hip && build/pcie_probe
// Mechanisms compared, pinned host memory throughout:
//   dma1 / dma2 / dma4   hipMemcpyAsync on 1 / 2 / 4 streams (the runtime's SDMA engines)
//   kern                  a copy KERNEL reading the mapped host buffer (shader blit: every CU issues PCIe reads)
//   pageable              hipMemcpy from malloc'ed memory (what a naive caller pays)
// for H2D, D2H and both directions at once, at one-frame (2 MB), chunk (100 MB) and large (512 MB) sizes.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

__global__ void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  const size_t big = (size_t)512 << 20;
  uint8_t *h_in, *h_out, *d_a, *d_b;
  CK(hipSetDevice(0));
  CK(hipHostMalloc((void**)&h_in, big, hipHostMallocDefault));
  CK(hipHostMalloc((void**)&h_out, big, hipHostMallocDefault));
  CK(hipMalloc((void**)&d_a, big));
  CK(hipMalloc((void**)&d_b, big));
  memset(h_in, 0x5a, big);
  memset(h_out, 0, big);
  uint8_t* pageable = (uint8_t*)malloc(big);
  memset(pageable, 0x33, big);
  void *dh_in = nullptr, *dh_out = nullptr;
  CK(hipHostGetDevicePointer(&dh_in, h_in, 0));
  CK(hipHostGetDevicePointer(&dh_out, h_out, 0));
  hipStream_t st[8];
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));

  auto time_it = [&](const char* name, size_t bytes, int reps, auto fn) {
    fn();
    CK(hipDeviceSynchronize());
    const double t0 = now();
    for (int r = 0; r < reps; ++r) fn();
    CK(hipDeviceSynchronize());
    const double dt = (now() - t0) / reps;
    printf("  %-34s %8.1f MB  %8.3f ms  %7.2f GB/s\n", name, bytes / 1e6, dt * 1e3, bytes / dt / 1e9);
    fflush(stdout);
    return bytes / dt / 1e9;
  };

  const size_t sizes[] = {(size_t)1920 * 1080, (size_t)50 * 1920 * 1080, big};
  for (size_t bytes : sizes) {
    const int reps = bytes < ((size_t)8 << 20) ? 200 : (bytes < ((size_t)200 << 20) ? 20 : 6);
    printf("size %.1f MB\n", bytes / 1e6);
    for (int ns : {1, 2, 4}) {
      char nm[64];
      snprintf(nm, sizeof nm, "H2D dma x%d", ns);
      time_it(nm, bytes, reps, [&] {
        const size_t part = (bytes / ns + 255) & ~(size_t)255;
        for (int s = 0; s < ns; ++s) {
          const size_t o = s * part, n = o >= bytes ? 0 : (bytes - o < part ? bytes - o : part);
          if (n) CK(hipMemcpyAsync(d_a + o, h_in + o, n, hipMemcpyHostToDevice, st[s]));
        }
      });
    }
    for (int blocks : {256, 1024, 4096}) {
      char nm[64];
      snprintf(nm, sizeof nm, "H2D kernel %d wg", blocks);
      time_it(nm, bytes, reps, [&] { copy_kernel<<<blocks, 256, 0, st[0]>>>((const uint4*)dh_in, (uint4*)d_a, bytes / 16); });
    }
    for (int ns : {1, 2}) {
      char nm[64];
      snprintf(nm, sizeof nm, "D2H dma x%d", ns);
      time_it(nm, bytes, reps, [&] {
        const size_t part = (bytes / ns + 255) & ~(size_t)255;
        for (int s = 0; s < ns; ++s) {
          const size_t o = s * part, n = o >= bytes ? 0 : (bytes - o < part ? bytes - o : part);
          if (n) CK(hipMemcpyAsync(h_out + o, d_b + o, n, hipMemcpyDeviceToHost, st[s]));
        }
      });
    }
    time_it("D2H kernel 1024 wg", bytes, reps,
            [&] { copy_kernel<<<1024, 256, 0, st[1]>>>((const uint4*)d_b, (uint4*)dh_out, bytes / 16); });
    time_it("H2D dma + D2H dma (bytes each way)", bytes, reps, [&] {
      CK(hipMemcpyAsync(d_a, h_in, bytes, hipMemcpyHostToDevice, st[0]));
      CK(hipMemcpyAsync(h_out, d_b, bytes, hipMemcpyDeviceToHost, st[1]));
    });
    time_it("H2D kernel + D2H dma (each way)", bytes, reps, [&] {
      copy_kernel<<<1024, 256, 0, st[0]>>>((const uint4*)dh_in, (uint4*)d_a, bytes / 16);
      CK(hipMemcpyAsync(h_out, d_b, bytes, hipMemcpyDeviceToHost, st[1]));
    });
    if (bytes <= ((size_t)200 << 20))
      time_it("H2D pageable hipMemcpy", bytes, reps < 10 ? reps : 10,
              [&] { CK(hipMemcpy(d_a, pageable, bytes, hipMemcpyHostToDevice)); });
  }
  // host memcpy rate into pinned staging (what a submit-by-copy entry pays per producer thread)
  {
    const size_t bytes = (size_t)100 << 20;
    const double t0 = now();
    for (int r = 0; r < 5; ++r) memcpy(h_in, pageable, bytes);
    const double dt = (now() - t0) / 5;
    printf("host memcpy pageable -> pinned, 1 thread: %.2f GB/s\n", bytes / dt / 1e9);
  }
  return 0;
}
