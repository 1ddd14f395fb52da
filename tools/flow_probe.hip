// Wall-clock stamps of the diagonal workgroups of the single-launch factorisation (perf probe, not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -Iinclude tools/flow_probe.hip -o build/flow_probe -ldl -lrt
// Prints, per diagonal block j (times in us): how long the block's accumulators were ready before M_{j-1} arrived (slack),
// M fetch, X = P M^T, diagonal update + publish, potf2 + inverse, publishing M_j, and the step T_j - T_{j-1}.
#define GH_CHOL_PROBE 1
#include "../gslam_amd/csrc/chol.hip"
#include "../gslam_amd/csrc/ctx.hip"

#include <vector>

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 3000, lda = (n + 1 + 15) & ~15;
  std::vector<double> A((size_t)n * lda, 0.0);
  for (int c = 0; c < n; ++c) {
    for (int r = c; r < n; ++r) A[(size_t)c * lda + r] = (r == c) ? 80.0 : 1.0 / (1 + r - c);
    A[(size_t)c * lda + n] = 1.0 + 0.001 * c;  // right-hand-side row
  }
  gh_ctx* ctx = nullptr;
  if (gh_ctx_create(0, &ctx) != GH_OK) return 1;
  double *dA, *dM, *dX;
  int* dinfo;
  unsigned* dflow;
  const size_t words = gh_potrf_flow_words(ctx, n, 1);
  if (!words) { printf("shape not eligible\n"); return 1; }
  hipMalloc(&dA, A.size() * 8);
  hipMalloc(&dM, (size_t)((n + 63) / 64) * 4096 * 8);
  hipMalloc(&dX, (size_t)2 * 64 * (n + 1) * 8);
  hipMalloc(&dinfo, 4);
  hipMalloc(&dflow, words * 4);
  const int nb = (n + 63) / 64;
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, ctx->stream);
    if (gh_potrf_dev_impl(ctx, dA, n, lda, dinfo, 1, dM, dX, dflow, false, false) != GH_OK) { printf("launch failed: %s\n", ctx->last_error.c_str()); return 1; }
    hipEventRecord(e1, ctx->stream);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    int info = -1;
    hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost);
    std::vector<long long> st(128 * 8);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_flow_trace), st.size() * 8);
    std::vector<long long> cy(128 * 8);
    hipMemcpyFromSymbol(cy.data(), HIP_SYMBOL(g_flow_cycles), cy.size() * 8);
    if (rep == 2 && nb > 10)
      printf("shader clock during potf2 of block 10: %.0f MHz; over blocks 2..%d: %.0f MHz\n",
             (cy[10 * 8 + 6] - cy[10 * 8 + 5]) / ((st[10 * 8 + 6] - st[10 * 8 + 5]) * 0.01),
             nb - 1, (cy[(nb - 1) * 8 + 7] - cy[2 * 8 + 0]) / ((st[(nb - 1) * 8 + 7] - st[2 * 8 + 0]) * 0.01));
    printf("rep %d: n %d info %d  %.1f us (memset + launch)\n", rep, n, info, ms * 1e3);
    if (rep < 2) continue;
    auto us = [](long long d) { return d * 0.01; };
    const long long t0 = st[0];
    printf("  j    wait   trsm    upd+pub  potf2  stores   T_j      step   (us; wait = for the row's accumulators)\n");
    double sum[6] = {0, 0, 0, 0, 0, 0};
    for (int j = 0; j < nb; ++j) {
      const long long* s = &st[j * 8];
      const double step = j ? us(s[7] - st[(j - 1) * 8 + 7]) : us(s[7] - t0);
      if (j >= 1) {
        printf("%3d %7.2f %7.2f %7.2f %7.2f %6.2f %8.1f %7.2f\n", j, us(s[1] - s[0]), us(s[4] - s[3]), us(s[5] - s[4]),
               us(s[6] - s[5]), us(s[7] - s[6]), us(s[7] - t0), step);
        if (j >= 2) { sum[0] += us(s[1] - s[0]); sum[1] += us(s[4] - s[3]); sum[2] += us(s[5] - s[4]); sum[3] += us(s[6] - s[5]); sum[4] += us(s[7] - s[6]); sum[5] += step; }
      } else {
        printf("%3d %7.2f    -       -    %7.2f %6.2f %8.1f %7.2f\n", j, us(s[1] - s[0]), us(s[6] - s[5]), us(s[7] - s[6]), us(s[7] - t0), step);
      }
    }
    {  // the dependency loop: M_j published -> worker tile (j+2, j) -> accumulators of row j+2 -> chain step j+2
      std::vector<long long> lp(128 * 8);
      hipMemcpyFromSymbol(lp.data(), HIP_SYMBOL(g_flow_loop), lp.size() * 8);
      double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      int cntl = 0;
      for (int j = 4; j + 2 < nb; ++j) {
        const long long* l = &lp[j * 8];
        const long long tj = st[j * 8 + 6];  // end of potf2 of block j
        acc[0] += us(l[0] - tj); acc[1] += us(l[1] - tj); acc[2] += us(l[2] - tj); acc[3] += us(l[3] - tj);
        acc[4] += us(l[4] - tj); acc[5] += us(l[5] - tj); acc[6] += us(l[6] - tj);
        acc[7] += us(st[(j + 2) * 8 + 1] - tj);
        ++cntl;
      }
      {
        std::vector<long long> pr(64);
        hipMemcpyFromSymbol(pr.data(), HIP_SYMBOL(g_probe), pr.size() * 8);
        static const char* nm[] = {"panel 0", "update", "panel 1 | inv 0", "update", "panel 2 | inv 1, T", "update",
                                   "panel 3 | inv 2, 3, X, T64, poll", "request", "X32, X64 up, W", "X64 low"};
        printf("potf2 phases, mean over the blocks (us):");
        for (int e = 0; e < 10; ++e) printf("  %s %.2f", nm[e], pr[41 + e] * 0.01 / nb / 3);
        printf("\n   within the last-panel phase, us after its start: wave 0 pivots done %.2f, wave 1 inverse done %.2f, wave 3 %.2f, wave 2 %.2f, wave 7 %.2f, wave 5 after side %.2f, wave 6 after side %.2f\n",
               pr[56] * 0.01 / nb / 3, pr[57] * 0.01 / nb / 3, pr[58] * 0.01 / nb / 3, pr[59] * 0.01 / nb / 3, pr[60] * 0.01 / nb / 3, pr[61] * 0.01 / nb / 3, pr[62] * 0.01 / nb / 3);
        if (pr[12] > 0)
          printf("   the last worker of the last row (%lld tiles): %.2f us per pure update column (mean over %lld columns), %.2f us of them between having its rows of L[i,k] and the end of the column; next column's rows prefetched %.0f%%, first tile %.0f%%\n", pr[13],
                 pr[10] * 0.01 / 3 / pr[12], pr[12], pr[15] * 0.01 / 3 / (pr[12] + 1), pr[16] * 100.0 / 3 / (pr[12] + 1), pr[17] * 100.0 / 3 / (pr[12] + 1));
        long long zero[64] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(g_probe), zero, sizeof(zero));
      }
      if (cntl) {
        double dw = 0, fa = 0, iv = 0;
        for (int j = 4; j + 2 < nb; ++j) {
          dw += us(st[j * 8 + 2] - st[j * 8 + 5]);
          fa += us(lp[j * 8 + 7] - st[j * 8 + 2]);
          iv += us(st[j * 8 + 6] - lp[j * 8 + 7]);
        }
        printf("inside the potf2 phase, mean us: D to LDS + barrier %.2f, 64 pivots + rank-16 updates %.2f, prefetch + inverse %.2f\n",
               dw / cntl, fa / cntl, iv / cntl);
      }
      if (cntl)
        printf("loop, mean us after the end of potf2(j): M stores issued %.2f, M published %.2f, worker saw it %.2f, tile (j+2, j) "
               "published %.2f, row j+2 saw its column-j tiles %.2f, added them %.2f, handed over %.2f, chain has them %.2f\n",
               acc[0] / cntl, acc[1] / cntl, acc[2] / cntl, acc[3] / cntl, acc[4] / cntl, acc[5] / cntl, acc[6] / cntl, acc[7] / cntl);
    }
    const int cnt = nb - 2;
    if (cnt > 0)
      printf("mean over j >= 2: wait %.2f trsm %.2f upd+pub %.2f potf2 %.2f stores %.2f step %.2f us\n", sum[0] / cnt, sum[1] / cnt,
             sum[2] / cnt, sum[3] / cnt, sum[4] / cnt, sum[5] / cnt);
  }
  return 0;
}
