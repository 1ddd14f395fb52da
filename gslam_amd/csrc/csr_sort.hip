// Index lists (CSR) of a large graph built on the GPU: items sorted by key, ascending item index inside a key -- what
// build_csr (ba.hip) does on the host with a counting sort.  For the 6 M observations of BASELINE configs[2] (C5) the host
// teams take ~60 ms per solve, a stable LSD radix sort of (key, item) pairs on the device ~1 ms plus the transfers.
// rocPRIM is the sort (plumbing, like the runtime's memcpy); everything that touches it is in this file.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace {

__global__ __launch_bounds__(256) void csr_prepare_kernel(const int32_t* __restrict__ key, int n, int n_keys, int32_t* __restrict__ item,
                                                          int32_t* __restrict__ count, int* __restrict__ bad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  item[i] = i;
  const int k = key[i];
  if (k < 0 || k >= n_keys) {
    *bad = 1;
    return;
  }
  atomicAdd(&count[k], 1);
}

}  // namespace

// keys_host: n_items keys in [0, n_keys); start_host: n_keys + 1 offsets; list_host: n_items item indices, grouped by key in
// ascending key order, ascending inside a key.  Device memory comes from (and goes back to) hipMalloc: a setup-time call.
gh_status gh_csr_build_dev(gh_ctx* ctx, const int32_t* keys_host, int n_items, int n_keys, int32_t* start_host, int32_t* list_host) {
  if (n_items <= 0 || n_keys <= 0) return gh_set_error(ctx, GH_ERR_ARG, "gh_csr_build_dev: empty input");
  int bits = 1;
  while (bits < 31 && (1ll << bits) < n_keys) ++bits;
  const size_t N = (size_t)n_items, K = (size_t)n_keys + 1;
  size_t sort_bytes = 0, scan_bytes = 0;
  int32_t* null_i = nullptr;
  if (rocprim::radix_sort_pairs(nullptr, sort_bytes, null_i, null_i, null_i, null_i, N, 0, bits, ctx->stream) != hipSuccess ||
      rocprim::exclusive_scan(nullptr, scan_bytes, null_i, null_i, 0, K, rocprim::plus<int32_t>(), ctx->stream) != hipSuccess)
    return gh_set_error(ctx, GH_ERR_HIP, "gh_csr_build_dev: rocPRIM size query failed");
  const size_t tmp_bytes = (std::max(sort_bytes, scan_bytes) + 255) & ~(size_t)255;
  char* base = nullptr;
  const size_t total = 4 * N * 4 + 2 * K * 4 + tmp_bytes + 1024;
  if (hipMalloc((void**)&base, total) != hipSuccess) return gh_set_error(ctx, GH_ERR_NOMEM, "gh_csr_build_dev: hipMalloc(%zu)", total);
  int32_t* d_key = (int32_t*)base;
  int32_t* d_key_out = d_key + N;
  int32_t* d_item = d_key_out + N;
  int32_t* d_item_out = d_item + N;
  int32_t* d_count = d_item_out + N;
  int32_t* d_start = d_count + K;
  int* d_bad = (int*)(d_start + K);
  void* d_tmp = (char*)base + ((4 * N * 4 + 2 * K * 4 + 256 + 255) & ~(size_t)255);
  auto fail = [&](const char* what) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(base);
    return gh_set_error(ctx, GH_ERR_HIP, "gh_csr_build_dev: %s", what);
  };
  if (hipMemcpyAsync(d_key, keys_host, N * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return fail("upload");
  if (hipMemsetAsync(d_count, 0, 2 * K * 4 + 256, ctx->stream) != hipSuccess) return fail("memset");
  hipLaunchKernelGGL(csr_prepare_kernel, dim3(gh_div_up(n_items, 256)), dim3(256), 0, ctx->stream, (const int32_t*)d_key, n_items, n_keys,
                     d_item, d_count, d_bad);
  size_t tb = tmp_bytes;
  if (rocprim::exclusive_scan(d_tmp, tb, d_count, d_start, 0, K, rocprim::plus<int32_t>(), ctx->stream) != hipSuccess) return fail("scan");
  tb = tmp_bytes;
  if (rocprim::radix_sort_pairs(d_tmp, tb, d_key, d_key_out, d_item, d_item_out, N, 0, bits, ctx->stream) != hipSuccess)
    return fail("sort");
  int bad = 0;
  if (hipMemcpyAsync(list_host, d_item_out, N * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipMemcpyAsync(start_host, d_start, K * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess)
    return fail("download");
  (void)hipFree(base);
  if (bad) return gh_set_error(ctx, GH_ERR_ARG, "gh_csr_build_dev: a key is outside [0, %d)", n_keys);
  return GH_OK;
}
