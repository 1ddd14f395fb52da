// What does FETCH_SIZE count for orb_describe's access pattern?  
// The guide's "FETCH_SIZE reports half the bytes" was calibrated on wide 16 B / lane streaming reads; orb_describe reads 33 rows
// of 36 bytes (nine dwords per lane, 4-byte aligned, one row per lane) around every keypoint.  This tool issues exactly that
// pattern at known random places of a buffer far larger than L2 + Infinity Cache and prints the bytes a memory system would have
// to move at 32 / 64 / 128-byte granularity (unique sectors per patch: neighbouring patches are random, they share nothing).
// Run it under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (and TCC_EA0_RDREQ_sum / TCC_EA0_RDREQ_32B_sum): whichever granularity
// FETCH_SIZE matches is what the counter means for this pattern.  A second kernel streams the same buffer with 16 B / lane loads:
// the pattern the guide's factor of two was calibrated on.
//   hipcc --offload-arch=gfx950 -O3 -o build/fetch_calib tools/fetch_calib.hip && build/fetch_calib
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kPitch = 1920, kRows = 33, kRowDw = 9;
struct __attribute__((packed, aligned(4))) RowN { uint32_t w[kRowDw]; };

// one wave per patch: lane r < 33 reads the nine dwords of row r at off[patch] + r * pitch
__global__ __launch_bounds__(256) void patch_rows_kernel(const uint8_t* __restrict__ buf, const uint64_t* __restrict__ off, int n,
                                                         uint32_t* __restrict__ out) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + wv;
  if (p >= n) return;
  const uint64_t o = off[p];
  const RowN row = *reinterpret_cast<const RowN*>(buf + o + (uint64_t)(lane < kRows ? lane : kRows - 1) * kPitch);
  uint32_t s = 0;
#pragma unroll
  for (int c = 0; c < kRowDw; ++c) s ^= row.w[c];
  if (s == 0x12345u) out[p] = s;  // (never: keeps the loads)
}

__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ buf, size_t n16, uint32_t* __restrict__ out) {
  uint32_t s = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256ull) {
    const uint4 v = buf[i];
    s ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (s == 0x12345u) out[0] = s;
}

int main(int argc, char** argv) {
  const size_t bytes = 8ull << 30;  // 8 GB: 30 x (L2 + Infinity Cache)
  const int n = argc > 1 ? atoi(argv[1]) : 2000000;  // patches per launch = keypoints of one C2 step
  uint8_t* buf;
  CK(hipMalloc(&buf, bytes));
  CK(hipMemset(buf, 1, bytes));
  std::vector<uint64_t> off((size_t)n);
  uint64_t st = 0x9E3779B97F4A7C15ull;
  double need[3] = {0, 0, 0};
  const int gran[3] = {32, 64, 128};
  for (int p = 0; p < n; ++p) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    const uint64_t o = ((st >> 11) % (bytes - (uint64_t)(kRows + 1) * kPitch)) & ~3ull;  // 4-byte aligned, as (x - 15) & ~3
    off[p] = o;
    for (int g = 0; g < 3; ++g)  // (rows are 1920 bytes apart: they share no sector)
      for (int r = 0; r < kRows; ++r) {
        const uint64_t a0 = o + (uint64_t)r * kPitch, a1 = a0 + 4 * kRowDw - 1;
        need[g] += (double)((a1 / gran[g] - a0 / gran[g] + 1) * gran[g]);
      }
  }
  uint64_t* d_off;
  uint32_t* d_out;
  CK(hipMalloc(&d_off, (size_t)n * 8));
  CK(hipMalloc(&d_out, (size_t)n * 4));
  CK(hipMemcpy(d_off, off.data(), (size_t)n * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    patch_rows_kernel<<<(n + 3) / 4, 256>>>(buf, d_off, n, d_out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("patch_rows_kernel: %d patches of 33 rows x 36 B (useful %.3f GB) in %.3f ms; bytes at 32 / 64 / 128 B granularity: %.3f / %.3f / %.3f GB "
           "(%.2f / %.2f / %.2f TB/s)\n", n, n * 33.0 * 36 / 1e9, ms, need[0] / 1e9, need[1] / 1e9, need[2] / 1e9, need[0] / ms / 1e9,
           need[1] / ms / 1e9, need[2] / ms / 1e9);
  }
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    stream_kernel<<<256 * 16, 256>>>(reinterpret_cast<const uint4*>(buf), bytes / 16, d_out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("stream_kernel: %.3f GB with 16 B / lane in %.3f ms (%.2f TB/s)\n", bytes / 1e9, ms, bytes / ms / 1e9);
  }
  return 0;
}
