// Probe (GPU box): what v_mfma_f32_16x16x32_f16 returns for the operands orb_describe's MFMA blur builds -- prints, for every
// (mb, nb, lane, reg), whether D equals 2^21 + H[row][col] of the expected (row, col), and if not which (row, col) it matches.
// build: hipcc --offload-arch=gfx950 -O2 -o build/probe/mfma_blur_probe tools/mfma_blur_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

__global__ void probe(const uint8_t* patch /* 37 x 36 */, const uint32_t* btab /* 64 x 4 */, float* out /* [2][4][64][4] */, uint8_t* blur_out /* 1024 */) {
  __shared__ __attribute__((aligned(16))) uint8_t sp[37 * 36 + 28];
  __shared__ __attribute__((aligned(16))) uint8_t bl[1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < 37 * 36; i += 64) sp[i] = patch[i];
  __syncthreads();
  const int n16 = lane & 15, q4 = lane >> 4;
  const uint4 bw = *reinterpret_cast<const uint4*>(btab + 4 * lane);
  const f16x8 bfrag = __builtin_bit_cast(f16x8, bw);
  const uint32_t* arow = reinterpret_cast<const uint32_t*>(sp) + (7 * (n16 >> 2) + (lane & 3)) * 9 + 2 * q4;
  const uint32_t k64 = 0x64646464u;
  uint32_t hs[2][16];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const uint32_t d0 = arow[4 * mb * 9 + 4 * nb], d1 = arow[4 * mb * 9 + 4 * nb + 1];
      uint4 af;
      af.x = __builtin_amdgcn_perm(k64, d0, 0x04010400u);
      af.y = __builtin_amdgcn_perm(k64, d0, 0x04030402u);
      af.z = __builtin_amdgcn_perm(k64, d1, 0x04010400u);
      af.w = __builtin_amdgcn_perm(k64, d1, 0x04030402u);
      const f32x4 c0 = {0.0f, 0.0f, 0.0f, 0.0f};
      const f32x4 dd = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af), bfrag, c0, 0, 0, 0);
      float* o = out + ((nb * 4 + mb) * 64 + lane) * 4;
      o[0] = dd[0]; o[1] = dd[1]; o[2] = dd[2]; o[3] = dd[3];
      hs[nb][4 * mb + 0] = __float_as_uint(dd[0]);
      hs[nb][4 * mb + 1] = __float_as_uint(dd[1]);
      hs[nb][4 * mb + 2] = __float_as_uint(dd[2]);
      hs[nb][4 * mb + 3] = __float_as_uint(dd[3]);
    }
  constexpr uint32_t g4[7] = {144, 268, 391, 442, 391, 268, 144};
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    uint32_t o[8];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      uint32_t acc = 1u << 23;
#pragma unroll
      for (int t = 0; t < 7; ++t) acc = mad_u24(hs[nb][i + t], g4[t], acc);
      o[i] = acc;
    }
    o[7] = 0u;
    uint2 pk;
    pk.x = __builtin_amdgcn_perm(__builtin_amdgcn_perm(o[3], o[2], 0x0c0c0703u), __builtin_amdgcn_perm(o[1], o[0], 0x0c0c0703u), 0x05040100u);
    pk.y = __builtin_amdgcn_perm(__builtin_amdgcn_perm(o[7], o[6], 0x0c0c0703u), __builtin_amdgcn_perm(o[5], o[4], 0x0c0c0703u), 0x05040100u);
    *reinterpret_cast<uint2*>(bl + (16 * nb + n16) * 32 + 8 * q4) = pk;
  }
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) blur_out[i] = bl[i];
}

static uint32_t f16_bits(int v) {
  if (v == 0) return 0;
  int e = 0;
  while ((v >> (e + 1)) != 0) ++e;
  return (uint32_t)(((e + 15) << 10) | ((v << (10 - e)) & 0x3FF));
}

int main() {
  std::vector<uint8_t> patch(37 * 36);
  srand(7);
  for (auto& b : patch) b = (uint8_t)(rand() & 255);
  const int g[7] = {144, 268, 391, 442, 391, 268, 144};
  std::vector<uint32_t> bt(256, 0);
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < 8; ++e) {
      const int d = 8 * (l >> 4) + e - (l & 15);
      const uint32_t h = (d >= 0 && d <= 6) ? f16_bits(g[d]) : 0u;
      bt[4 * l + (e >> 1)] |= h << (16 * (e & 1));
    }
  uint8_t* dp; uint32_t* db; float* dout;
  hipMalloc(&dp, patch.size()); hipMalloc(&db, 1024); hipMalloc(&dout, 2 * 4 * 64 * 4 * 4);
  hipMemcpy(dp, patch.data(), patch.size(), hipMemcpyHostToDevice);
  hipMemcpy(db, bt.data(), 1024, hipMemcpyHostToDevice);
  uint8_t* dbl; hipMalloc(&dbl, 1024);
  probe<<<1, 64>>>(dp, db, dout, dbl);
  std::vector<uint8_t> blur(1024);
  hipMemcpy(blur.data(), dbl, 1024, hipMemcpyDeviceToHost);
  std::vector<float> out(2 * 4 * 64 * 4);
  hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
  auto H = [&](int r, int c) { long s = 0; for (int k = 0; k < 7; ++k) s += (long)g[k] * patch[r * 36 + c + k]; return s; };
  int bad = 0, shown = 0;
  for (int nb = 0; nb < 2; ++nb)
    for (int mb = 0; mb < 4; ++mb)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
          const int row = 7 * (l >> 4) + 4 * mb + j, col = 16 * nb + (l & 15);
          if (row > 36 || col + 6 > 35) continue;
          const float got = out[((nb * 4 + mb) * 64 + l) * 4 + j];
          const double want = 2097152.0 + (double)H(row, col);
          if ((double)got != want) {
            ++bad;
            if (shown < 12) {
              ++shown;
              printf("nb %d mb %d lane %2d reg %d: got %.2f want %.2f (row %d col %d)", nb, mb, l, j, got, want, row, col);
              for (int r = 0; r < 37; ++r)
                for (int c = 0; c + 6 < 36; ++c)
                  if ((double)got == 2097152.0 + (double)H(r, c)) printf("  == H[%d][%d]", r, c);
              printf("\n");
            }
          }
        }
  printf("mfma_blur_probe: %d mismatches in the h-pass\n", bad);
  int bad2 = 0;
  for (int r = 0; r < 27; ++r)
    for (int c = 0; c < 27; ++c) {
      long sv = 0;
      for (int t = 0; t < 7; ++t) sv += (long)g[t] * H(r + t, c);
      const int want = (int)((sv + (1l << 21)) >> 22), got = blur[c * 32 + 8 * (r / 7) + r % 7];
      if (want != got) {
        if (bad2 < 12) printf("blur[%d][%d]: got %d want %d\n", r, c, got, want);
        ++bad2;
      }
    }
  printf("mfma_blur_probe: %d mismatches in the blurred patch\n", bad2);
  bad += bad2;
  return bad != 0;
}
