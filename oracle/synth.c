/*
 * synth.c — deterministic synthetic gray frames (integer procedural texture).  TEST INFRASTRUCTURE:
 * the CPU twin of gh_synth_frames_dev (gslam_amd/csrc/synth.hip); both must agree bit for bit.
 *
 * No reference code exists for this (GSLAM reads datasets from disk); the spec is:
 * the image is cut into 128x128 tiles; tile (tx,ty) of frame seed S gets a base level and 12 shapes
 * (rectangles, diamonds, discs) from splitmix64-style hashing, clipped to the tile; plus +-4 of
 * per-pixel hash noise.  Everything is unsigned integer arithmetic.
 */
#include <stdint.h>
#include <stdlib.h>

#define SYNTH_TILE 128
#define SYNTH_SHAPES 12

static inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

typedef struct {
  int type, cx, cy, hw, hh, delta;
} synth_shape;

static int tile_shapes(uint32_t seed, int tx, int ty, synth_shape* sh) {
  uint64_t key = ((uint64_t)seed << 32) ^ ((uint64_t)(uint32_t)ty << 16) ^ (uint64_t)(uint32_t)tx;
  uint64_t s = mix64(key + 0x9E3779B97F4A7C15ull);
  for (int k = 0; k < SYNTH_SHAPES; ++k) {
    uint64_t r = mix64(s + (uint64_t)(k + 1) * 0x9E3779B97F4A7C15ull);
    sh[k].type = (int)(r & 3);
    sh[k].cx = (int)((r >> 2) & 127);
    sh[k].cy = (int)((r >> 9) & 127);
    sh[k].hw = 3 + (int)((r >> 16) & 31);
    sh[k].hh = 3 + (int)((r >> 21) & 31);
    int d = (int)((r >> 26) & 127) - 64;
    sh[k].delta = d >= 0 ? d + 12 : d - 12;
  }
  return 96 + (int)(s & 63);
}

static inline int synth_noise(uint32_t x, uint32_t y, uint32_t seed) {
  uint32_t h = (x * 0x9E3779B1u) ^ (y * 0x85EBCA77u) ^ (seed * 0xC2B2AE3Du);
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 12;
  h *= 0x297A2D39u;
  h ^= h >> 15;
  return (int)(h & 7) - 4;
}

void oracle_synth_frame(uint8_t* out, int w, int h, int stride, uint32_t seed) {
  int ntx = (w + SYNTH_TILE - 1) / SYNTH_TILE, nty = (h + SYNTH_TILE - 1) / SYNTH_TILE;
  synth_shape sh[SYNTH_SHAPES];
  for (int ty = 0; ty < nty; ++ty)
    for (int tx = 0; tx < ntx; ++tx) {
      int base = tile_shapes(seed, tx, ty, sh);
      int x1 = (tx + 1) * SYNTH_TILE < w ? (tx + 1) * SYNTH_TILE : w;
      int y1 = (ty + 1) * SYNTH_TILE < h ? (ty + 1) * SYNTH_TILE : h;
      for (int y = ty * SYNTH_TILE; y < y1; ++y)
        for (int x = tx * SYNTH_TILE; x < x1; ++x) {
          int lx = x & (SYNTH_TILE - 1), ly = y & (SYNTH_TILE - 1);
          int v = base;
          for (int k = 0; k < SYNTH_SHAPES; ++k) {
            int dx = lx - sh[k].cx, dy = ly - sh[k].cy;
            int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
            int in;
            if (sh[k].type <= 1) in = adx <= sh[k].hw && ady <= sh[k].hh;
            else if (sh[k].type == 2) in = adx + ady <= sh[k].hw;
            else in = dx * dx + dy * dy <= sh[k].hw * sh[k].hw;
            if (in) v += sh[k].delta;
          }
          v += synth_noise((uint32_t)x, (uint32_t)y, seed);
          out[(size_t)y * stride + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}
