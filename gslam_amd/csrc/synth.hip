// Deterministic synthetic gray frames generated directly in HBM (test / bench input).
// Bit-identical twin of oracle/synth.c: 128x128 tiles, 12 hashed shapes per tile, +-4 hash noise.
// No reference counterpart (GSLAM reads datasets from disk; SURVEY.md 8d specifies synthetic input).
#include "common.h"

namespace {

constexpr int kTile = 128;
constexpr int kShapes = 12;

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct Shape {
  int type, cx, cy, hw, hh, delta;
};

__global__ __launch_bounds__(256) void synth_kernel(uint8_t* __restrict__ out, int w, int h, int row_stride,
                                                    size_t frame_stride, int first_frame, uint32_t base_seed) {
  __shared__ Shape sh[kShapes];
  __shared__ int s_base;
  const int tx = blockIdx.x, ty = blockIdx.y, f = blockIdx.z;
  const uint32_t seed = base_seed + (uint32_t)(first_frame + f);
  const uint64_t key = ((uint64_t)seed << 32) ^ ((uint64_t)(uint32_t)ty << 16) ^ (uint64_t)(uint32_t)tx;
  const uint64_t s = mix64(key + 0x9E3779B97F4A7C15ull);
  if (threadIdx.x < kShapes) {
    const int k = threadIdx.x;
    uint64_t r = mix64(s + (uint64_t)(k + 1) * 0x9E3779B97F4A7C15ull);
    Shape q;
    q.type = (int)(r & 3);
    q.cx = (int)((r >> 2) & 127);
    q.cy = (int)((r >> 9) & 127);
    q.hw = 3 + (int)((r >> 16) & 31);
    q.hh = 3 + (int)((r >> 21) & 31);
    int d = (int)((r >> 26) & 127) - 64;
    q.delta = d >= 0 ? d + 12 : d - 12;
    sh[k] = q;
  }
  if (threadIdx.x == 0) s_base = 96 + (int)(s & 63);
  __syncthreads();
  uint8_t* frame = out + (size_t)f * frame_stride;
  for (int p = threadIdx.x; p < kTile * kTile; p += 256) {
    const int lx = p & (kTile - 1), ly = p >> 7;
    const int x = tx * kTile + lx, y = ty * kTile + ly;
    if (x >= w || y >= h) continue;
    int v = s_base;
#pragma unroll
    for (int k = 0; k < kShapes; ++k) {
      const int dx = lx - sh[k].cx, dy = ly - sh[k].cy;
      const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
      bool in;
      if (sh[k].type <= 1) in = adx <= sh[k].hw && ady <= sh[k].hh;
      else if (sh[k].type == 2) in = adx + ady <= sh[k].hw;
      else in = dx * dx + dy * dy <= sh[k].hw * sh[k].hw;
      if (in) v += sh[k].delta;
    }
    uint32_t hsh = ((uint32_t)x * 0x9E3779B1u) ^ ((uint32_t)y * 0x85EBCA77u) ^ (seed * 0xC2B2AE3Du);
    hsh ^= hsh >> 15;
    hsh *= 0x2C1B3C6Du;
    hsh ^= hsh >> 12;
    hsh *= 0x297A2D39u;
    hsh ^= hsh >> 15;
    v += (int)(hsh & 7) - 4;
    frame[(size_t)y * row_stride + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

}  // namespace

extern "C" gh_status gh_synth_frames_dev(gh_ctx* ctx, uint8_t* gray_dev, int width, int height, int row_stride,
                                         size_t frame_stride, int first_frame, int n_frames, uint32_t base_seed) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, gray_dev && width > 0 && height > 0 && row_stride >= width && n_frames >= 0);
  if (n_frames == 0) return GH_OK;
  GH_CHECK_ARG(ctx, frame_stride >= (size_t)row_stride * height);
  for (int f0 = 0; f0 < n_frames; f0 += 65535) {
    int nf = n_frames - f0 < 65535 ? n_frames - f0 : 65535;
    dim3 grid(gh_div_up(width, kTile), gh_div_up(height, kTile), nf);
    GH_LAUNCH(ctx, "synth_frames", synth_kernel, grid, dim3(256), 0, gray_dev + (size_t)f0 * frame_stride, width,
              height, row_stride, frame_stride, first_frame + f0, base_seed);
  }
  return GH_OK;
}
