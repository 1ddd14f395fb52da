// Host-code sanitizer harness for gslam_amd/csrc/ba_order.hip (no GPU): random trajectory graphs -- in order, shuffled, with long-range
// observations, with the observation list in random order -- through gh_ba_camera_order under AddressSanitizer + UBSan.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -D__HIP_PLATFORM_AMD__ -Iinclude -Igslam_amd/csrc -I/opt/rocm/include \
//       -x c++ gslam_amd/csrc/ba_order.hip tools/order_asan.cpp -o build/order_asan -lpthread && ASAN_OPTIONS=detect_leaks=0 build/order_asan
// (tests/test_ba_order_sanitizers.py builds and runs it.)
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include <algorithm>
#include "gslam_hip.h"
int main() {
  std::mt19937 rng(7);
  for (int trial = 0; trial < 40; ++trial) {
    const int nc = 130 + rng() % 900, np = 200 + rng() % 20000, k = 2 + rng() % 6;
    std::vector<int32_t> oc, op;
    std::vector<int32_t> shuffle(nc);
    for (int i = 0; i < nc; ++i) shuffle[i] = i;
    if (trial & 1) std::shuffle(shuffle.begin(), shuffle.end(), rng);
    for (int p = 0; p < np; ++p) {
      const int home = rng() % nc;
      for (int j = 0; j < k; ++j) {
        int c = home - 12 + (int)(rng() % 25);
        if (trial % 5 == 0 && p % 997 == 0 && j == 0) c = rng() % nc;  // a few long-range observations
        c = c < 0 ? 0 : (c >= nc ? nc - 1 : c);
        oc.push_back(shuffle[c]);
        op.push_back(p);
      }
    }
    if (trial % 7 == 3) {  // observation list in random order
      for (size_t i = oc.size() - 1; i > 0; --i) { size_t j = rng() % (i + 1); std::swap(oc[i], oc[j]); std::swap(op[i], op[j]); }
    }
    gh_ba_problem pr{};
    pr.n_cams = nc; pr.n_points = np; pr.n_obs = (int)oc.size(); pr.obs_cam = oc.data(); pr.obs_point = op.data();
    std::vector<int32_t> perm(nc);
    int32_t nb = 0, span = 0, re = 0, nbp = 0;
    const gh_status st = gh_ba_camera_order(&pr, perm.data(), &nb, &span, &re, &nbp);
    std::vector<char> seen(nc, 0);
    for (int c : perm) { if (c < 0 || c >= nc || seen[c]) { printf("trial %d: not a permutation\n", trial); return 1; } seen[c] = 1; }
    printf("trial %2d: nc %4d np %5d k %d -> status %d border cams %3d border points %3d span %4d reordered %d\n", trial, nc, np, k, (int)st, nb, nbp, span, re);
  }
  return 0;
}
