/*
 * bf_oracle.c — CPU restatement of the brute-force Hamming matcher.  TEST INFRASTRUCTURE ONLY:
 * imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product.
 *
 * Follows, line by line:
 *   distance   GSLAM/core/Vocabulary.h:485-491   DistanceFactory::hamming32 — reinterpret the two
 *              32-byte rows as 4 little-endian uint64_t, XOR, popcount, sum (float-typed, exact).
 *   selection  GSLAM/core/Vocabulary.h:1712-1725 — best_d starts at FLT_MAX, `if (d < best_d)`:
 *              strict '<' so the FIRST minimum (lowest train index) wins ties.
 * Pinning: oracle/_ref/libgslam_ref.so compiles the reference's own hamming32 from
 * /root/reference/GSLAM/core/Vocabulary.h; tests/test_bf_oracle.py checks this file against it (live, where
 * /root/reference exists) and against tests/golden/bf_reference.npz generated from it (tools/gen_golden.py).
 */
#include <stdint.h>
#include <string.h>

/* software popcount, independent of -mpopcnt (matches std::bitset<64>::count) */
static inline int popcount64(uint64_t x) {
  x = x - ((x >> 1) & 0x5555555555555555ull);
  x = (x & 0x3333333333333333ull) + ((x >> 2) & 0x3333333333333333ull);
  x = (x + (x >> 4)) & 0x0f0f0f0f0f0f0f0full;
  return (int)((x * 0x0101010101010101ull) >> 56);
}

int oracle_hamming32(const uint8_t* a, const uint8_t* b) {
  uint64_t pa[4], pb[4];
  memcpy(pa, a, 32);
  memcpy(pb, b, 32);
#if defined(__POPCNT__)
  return __builtin_popcountll(pa[0] ^ pb[0]) + __builtin_popcountll(pa[1] ^ pb[1]) +
         __builtin_popcountll(pa[2] ^ pb[2]) + __builtin_popcountll(pa[3] ^ pb[3]);
#else
  return popcount64(pa[0] ^ pb[0]) + popcount64(pa[1] ^ pb[1]) + popcount64(pa[2] ^ pb[2]) +
         popcount64(pa[3] ^ pb[3]);
#endif
}

/* idx1 = first minimum; d2 = minimum over j != idx1 (again first-min; only the value is kept). */
void oracle_bf_match(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx1, uint16_t* d1,
                     uint16_t* d2) {
  for (int i = 0; i < nq; ++i) {
    int best_d = 1 << 30, best_j = -1, second_d = 1 << 30;
    const uint8_t* qi = q + (size_t)i * 32;
    for (int j = 0; j < nt; ++j) {
      int d = oracle_hamming32(qi, t + (size_t)j * 32);
      if (d < best_d) {
        second_d = best_d;
        best_d = d;
        best_j = j;
      } else if (d < second_d) {
        second_d = d;
      }
    }
    idx1[i] = best_j;
    d1[i] = best_j >= 0 ? (uint16_t)best_d : 65535;
    d2[i] = second_d < (1 << 30) ? (uint16_t)second_d : 65535;
  }
}

/* Descriptors of any width that is a multiple of 8 bytes -- GSLAM/core/Vocabulary.h:493-513: DistanceFactory::hamming64 (eight
 * uint64_t words) and hamming8x (bytes / 8 words: trailing bytes beyond a multiple of 8 are not compared) -- with the same
 * first-minimum rule.  Pinned like hamming32: oracle/ref_shim.cpp compiles the reference's own two functions
 * (tests/test_bf_oracle.py, tests/golden/bf_bytes_reference.npz). */
int oracle_hamming8x(const uint8_t* a, const uint8_t* b, int bytes) {
  int d = 0;
  for (int w = 0; w < bytes / 8; ++w) {
    uint64_t pa, pb;
    memcpy(&pa, a + 8 * w, 8);
    memcpy(&pb, b + 8 * w, 8);
    d += popcount64(pa ^ pb);
  }
  return d;
}

void oracle_bf_match_bytes(const uint8_t* q, int nq, const uint8_t* t, int nt, int bytes, int32_t* idx1, uint16_t* d1,
                           uint16_t* d2) {
  for (int i = 0; i < nq; ++i) {
    int best_d = 1 << 30, best_j = -1, second_d = 1 << 30;
    const uint8_t* qi = q + (size_t)i * bytes;
    for (int j = 0; j < nt; ++j) {
      int d = oracle_hamming8x(qi, t + (size_t)j * bytes, bytes);
      if (d < best_d) {
        second_d = best_d;
        best_d = d;
        best_j = j;
      } else if (d < second_d) {
        second_d = d;
      }
    }
    idx1[i] = best_j;
    d1[i] = best_j >= 0 ? (uint16_t)best_d : 65535;
    d2[i] = second_d < (1 << 30) ? (uint16_t)second_d : 65535;
  }
}

/* Multi-threaded variant for the timed CPU baseline (parallel over query rows). */
void oracle_bf_match_omp(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx1, uint16_t* d1,
                         uint16_t* d2, int threads) {
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int i = 0; i < nq; ++i) oracle_bf_match(q + (size_t)i * 32, 1, t, nt, idx1 + i, d1 + i, d2 + i);
}

void oracle_match_mask(const int32_t* idx1, const uint16_t* d1, const uint16_t* d2, int nq, const int32_t* back,
                       int nt, int max_dist, int ratio_num, int ratio_den, int cross_check, uint8_t* keep) {
  for (int i = 0; i < nq; ++i) {
    int j = idx1[i];
    int ok = j >= 0 && (int)d1[i] <= max_dist;
    if (ok && ratio_num > 0) ok = (int)d1[i] * ratio_den < ratio_num * (int)d2[i];
    if (ok && cross_check) ok = j < nt && back[j] == i;
    keep[i] = (uint8_t)(ok ? 1 : 0);
  }
}

/* Row-band restricted matcher (stereo): candidate j of query i iff |yq[i] - yt[j]| <= sizeq[i] * band_per_size
 * (fp32, one multiply), otherwise as oracle_bf_match.  No reference counterpart: the band rule restates ORB-SLAM's
 * stereo matcher (rows within +-2*scale) as SURVEY.md 8e prescribes ("same kernel + y-band mask"). */
void oracle_bf_match_band(const uint8_t* q, const float* yq, const float* sizeq, int nq, const uint8_t* t,
                          const float* yt, int nt, float band_per_size, int32_t* idx1, uint16_t* d1, uint16_t* d2) {
  for (int i = 0; i < nq; ++i) {
    int best_d = 1 << 30, best_j = -1, second_d = 1 << 30;
    const float band = sizeq[i] * band_per_size;
    for (int j = 0; j < nt; ++j) {
      float dy = yq[i] - yt[j];
      if (dy < 0) dy = -dy;
      if (!(dy <= band)) continue;
      int d = oracle_hamming32(q + (size_t)i * 32, t + (size_t)j * 32);
      if (d < best_d) {
        second_d = best_d;
        best_d = d;
        best_j = j;
      } else if (d < second_d) {
        second_d = d;
      }
    }
    idx1[i] = best_j;
    d1[i] = best_j >= 0 ? (uint16_t)best_d : 65535;
    d2[i] = second_d < (1 << 30) ? (uint16_t)second_d : 65535;
  }
}
