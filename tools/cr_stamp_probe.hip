// Wall-clock stamps inside the level-0 launches of the band solver (perf probe, not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude tools/cr_stamp_probe.hip -o build/cr_stamp_probe
#define GH_CR_PROBE 1
#include "../gslam_amd/csrc/chol_cr.hip"
#include "../gslam_amd/csrc/ctx.hip"

#include <vector>

int main() {
  const int n = 3000, hb = 149, lda = 3008;
  std::vector<double> A((size_t)n * lda, 0.0), b(n, 1.0);
  unsigned long long st = 12345;
  auto rnd = [&] { st = st * 6364136223846793005ull + 1442695040888963407ull; return ((st >> 33) & 0xFFFF) / 65536.0 - 0.5; };
  for (int c = 0; c < n; ++c)
    for (int r = c; r < n && r - c <= hb; ++r) A[(size_t)c * lda + r] = (r == c) ? 2.0 * hb : rnd();
  gh_ctx* ctx = nullptr;
  if (gh_ctx_create(0, &ctx) != GH_OK) return 1;
  double *dA, *db;
  hipMalloc(&dA, A.size() * 8);
  hipMalloc(&db, n * 8);
  for (int rep = 0; rep < 4; ++rep) {
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    int info = -1;
    if (gh_band_solve_dev(ctx, dA, n, lda, hb, db, &info) != GH_OK) { printf("failed: %s\n", gh_last_error(ctx)); return 1; }
    long long s[64];
    hipMemcpyFromSymbol(s, HIP_SYMBOL(g_cr_stamp), sizeof(s));
    auto us = [&](int a, int b2) { return (s[b2] - s[a]) * 0.01; };
    printf("rep %d info %d\n factor: loads+dump %.2f |", rep, info, us(0, 1));
    for (int k = 0; k < 3; ++k) {
      printf(" k%d potf2 %.2f store %.2f", k, us(1 + 4 * k, 2 + 4 * k), us(2 + 4 * k, 3 + 4 * k));
      if (k < 2) printf(" X %.2f trailing+dump %.2f |", us(3 + 4 * k, 4 + 4 * k), us(4 + 4 * k, 5 + 4 * k));
    }
    printf(" total %.2f us\n", us(0, 15));
    printf(" factor end -> panels start %.2f; panels: loads %.2f products %.2f store %.2f total %.2f\n", us(15, 16), us(16, 17), us(17, 18),
           us(18, 19), us(16, 19));
    printf(" panels end -> update start %.2f; update: loads issue %.2f phases %.2f store %.2f total %.2f\n", us(19, 24), us(24, 25),
           us(25, 26), us(26, 27), us(24, 27));
    printf(" back (level 0, workgroup 0): loads + products %.2f reduce + store %.2f total %.2f\n", us(32, 33), us(33, 36), us(32, 36));
  }
  return 0;
}
