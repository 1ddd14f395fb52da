// Cycle stamps inside the single-workgroup Cholesky kernels (perf probe, not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude tools/chol_probe.hip -o build/chol_probe
#define GH_CHOL_PROBE 1
#include "../gslam_amd/csrc/chol.hip"
#include "../gslam_amd/csrc/ctx.hip"

#include <vector>

int main() {
  const int n = 64, lda = 64;
  std::vector<double> A((size_t)n * lda, 0.0);
  for (int c = 0; c < n; ++c)
    for (int r = c; r < n; ++r) A[(size_t)c * lda + r] = (r == c) ? 80.0 : 1.0 / (1 + r - c);
  double *dA, *dM;
  int* dinfo;
  hipMalloc(&dA, A.size() * 8);
  hipMalloc(&dM, 4096 * 8);
  hipMalloc(&dinfo, 4);
  hipMemset(dinfo, 0, 4);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(potf2_inv_kernel, dim3(1), dim3(256), 0, 0, dA, lda, 0, 64, dinfo, dM);
    hipDeviceSynchronize();
    long long st[64];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_probe), sizeof(st));
    printf("rep %d: load %lld |", rep, st[1] - st[0]);
    for (int k = 0; k < 4; ++k)
      printf(" k%d: panel %lld update %lld |", k, st[2 + 2 * k] - (k ? st[3 + 2 * (k - 1)] : st[1]), st[3 + 2 * k] - st[2 + 2 * k]);
    printf(" inv16 %lld assemble %lld store %lld total %lld cycles\n", st[10] - st[9], st[20] - st[10], st[21] - st[20], st[21] - st[0]);
  }
  return 0;
}
