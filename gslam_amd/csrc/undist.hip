// Image undistortion (remap) on gfx950 — SURVEY.md 8 f2, the step in front of the pyramid.
// Bit-exact with GSLAM/core/Undistorter.h:206-348 (undistortFast / undistort) wherever the reference defines the
// output; pixels it leaves unwritten are 0 here.  The remap tables (:120-203) are built on the host by the
// reference's own camera code and uploaded once.  Pure gather: HBM/L2-bound, one thread per output pixel,
// tables read as 16-byte records, batched over frames.
#include "common.h"

struct gh_undist_plan {
  gh_ctx* ctx;
  int w_in, h_in, w_out, h_out;
  float* d_remapX;
  int32_t* d_fast;
  int4* d_idx;
  float4* d_coef;
};

namespace {

template <int C>
__global__ __launch_bounds__(256) void undistort_kernel(const uint8_t* __restrict__ img, size_t in_stride,
                                                        uint8_t* __restrict__ out, size_t out_stride, int n_out,
                                                        const float* __restrict__ remapX,
                                                        const int32_t* __restrict__ remapFast,
                                                        const int4* __restrict__ remapIdx,
                                                        const float4* __restrict__ remapCoef, int fast) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_out) return;
  const uint8_t* src = img + (size_t)blockIdx.y * in_stride;
  uint8_t* dst = out + (size_t)blockIdx.y * out_stride + (size_t)i * C;
  uint8_t v[C];
#pragma unroll
  for (int j = 0; j < C; ++j) v[j] = 0;
  if (fast) {
    const int f = remapFast[i];
    const bool ok = C == 1 ? f > 0 : (C == 3 ? f >= 0 : remapX[i] > 0.f);
    if (ok) {
#pragma unroll
      for (int j = 0; j < C; ++j) v[j] = src[(size_t)f * C + j];
    }
  } else {
    const float x = remapX[i];
    const bool ok = C == 1 ? !(x < 0.f) : x > 0.f;
    if (ok) {
      const int4 id = remapIdx[i];
      const float4 co = remapCoef[i];
#pragma unroll
      for (int j = 0; j < C; ++j) {
        float acc = __fmul_rn((float)src[(size_t)id.x * C + j], co.x);
        acc = __fadd_rn(acc, __fmul_rn((float)src[(size_t)id.y * C + j], co.y));
        acc = __fadd_rn(acc, __fmul_rn((float)src[(size_t)id.z * C + j], co.z));
        acc = __fadd_rn(acc, __fmul_rn((float)src[(size_t)id.w * C + j], co.w));
        v[j] = (uint8_t)(int)acc;  // truncating store, as the reference's float -> uchar conversion
      }
    }
  }
#pragma unroll
  for (int j = 0; j < C; ++j) dst[j] = v[j];
}

}  // namespace

extern "C" gh_status gh_undist_plan_create(gh_ctx* ctx, int w_in, int h_in, int w_out, int h_out,
                                           const float* remapX, const int32_t* remapFast, const int32_t* remapIdx,
                                           const float* remapCoef, gh_undist_plan** out) {
  if (!ctx || !out) return GH_ERR_ARG;
  GH_ENTER(ctx);
  *out = nullptr;
  GH_CHECK_ARG(ctx, w_in > 0 && h_in > 0 && w_out > 0 && h_out > 0 && remapX && remapFast && remapIdx && remapCoef);
  const size_t n = (size_t)w_out * h_out, n_in = (size_t)w_in * h_in;
  // The reference's own tables step one row / column past the image for source positions in the last row or column
  // (Undistorter.h:184-187; it reads out of bounds there).  Clamp those to the last pixel; reject anything else.
  std::vector<int32_t> idx(remapIdx, remapIdx + 4 * n);
  for (size_t i = 0; i < n; ++i) {
    GH_CHECK_ARG(ctx, remapFast[i] < (long long)n_in);
    for (int t = 0; t < 4; ++t) {
      GH_CHECK_ARG(ctx, idx[4 * i + t] >= 0 && (size_t)idx[4 * i + t] <= n_in + (size_t)w_in);
      if ((size_t)idx[4 * i + t] >= n_in) idx[4 * i + t] = (int32_t)(n_in - 1);
    }
  }
  gh_undist_plan* p = new (std::nothrow) gh_undist_plan();
  if (!p) return GH_ERR_NOMEM;
  *p = gh_undist_plan{ctx, w_in, h_in, w_out, h_out, nullptr, nullptr, nullptr, nullptr};
  gh_status st = gh_dev_alloc(ctx, n * 4, (void**)&p->d_remapX);
  if (st == GH_OK) st = gh_dev_alloc(ctx, n * 4, (void**)&p->d_fast);
  if (st == GH_OK) st = gh_dev_alloc(ctx, n * 16, (void**)&p->d_idx);
  if (st == GH_OK) st = gh_dev_alloc(ctx, n * 16, (void**)&p->d_coef);
  if (st == GH_OK) st = gh_dev_upload(ctx, p->d_remapX, remapX, n * 4);
  if (st == GH_OK) st = gh_dev_upload(ctx, p->d_fast, remapFast, n * 4);
  if (st == GH_OK) st = gh_dev_upload(ctx, p->d_idx, idx.data(), n * 16);
  if (st == GH_OK) st = gh_dev_upload(ctx, p->d_coef, remapCoef, n * 16);
  if (st != GH_OK) {
    void* ptrs[] = {p->d_remapX, p->d_fast, p->d_idx, p->d_coef};
    for (void* q : ptrs)
      if (q) hipFree(q);
    delete p;
    return st;
  }
  *out = p;
  return GH_OK;
}

extern "C" void gh_undist_plan_destroy(gh_undist_plan* p) {
  if (!p) return;
  GH_ENTER(p->ctx);
  hipStreamSynchronize(p->ctx->stream);
  hipFree(p->d_remapX);
  hipFree(p->d_fast);
  hipFree(p->d_idx);
  hipFree(p->d_coef);
  delete p;
}

extern "C" gh_status gh_undistort_dev(gh_undist_plan* p, const uint8_t* img_dev, int channels, int batch,
                                      size_t in_frame_stride, uint8_t* out_dev, size_t out_frame_stride, int fast) {
  if (!p) return GH_ERR_ARG;
  gh_ctx* ctx = p->ctx;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, img_dev && out_dev && batch >= 0 && batch <= 65535 && (channels == 1 || channels == 3 || channels == 4));
  GH_CHECK_ARG(ctx, fast || channels != 4);  // the reference's bilinear path is only defined for 1 and 3 channels
  if (batch == 0) return GH_OK;
  const int n = p->w_out * p->h_out;
  GH_CHECK_ARG(ctx, in_frame_stride >= (size_t)p->w_in * p->h_in * channels && out_frame_stride >= (size_t)n * channels);
  dim3 grid(gh_div_up(n, 256), batch);
  if (channels == 1)
    GH_LAUNCH(ctx, "undistort", undistort_kernel<1>, grid, dim3(256), 0, img_dev, in_frame_stride, out_dev,
              out_frame_stride, n, p->d_remapX, p->d_fast, p->d_idx, p->d_coef, fast);
  else if (channels == 3)
    GH_LAUNCH(ctx, "undistort", undistort_kernel<3>, grid, dim3(256), 0, img_dev, in_frame_stride, out_dev,
              out_frame_stride, n, p->d_remapX, p->d_fast, p->d_idx, p->d_coef, fast);
  else
    GH_LAUNCH(ctx, "undistort", undistort_kernel<4>, grid, dim3(256), 0, img_dev, in_frame_stride, out_dev,
              out_frame_stride, n, p->d_remapX, p->d_fast, p->d_idx, p->d_coef, fast);
  return GH_OK;
}

extern "C" gh_status gh_undistort_host(gh_undist_plan* p, const uint8_t* img, int channels, uint8_t* out, int fast) {
  if (!p) return GH_ERR_ARG;
  gh_ctx* ctx = p->ctx;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, img && out && (channels == 1 || channels == 3 || channels == 4));
  const size_t nin = (size_t)p->w_in * p->h_in * channels, nout = (size_t)p->w_out * p->h_out * channels;
  void* s = nullptr;
  const size_t a = (nin + 255) & ~(size_t)255;
  GH_TRY(gh_scratch(ctx, a + nout, &s));
  uint8_t* d_in = (uint8_t*)s;
  uint8_t* d_out = d_in + a;
  GH_HIP(ctx, hipMemcpyAsync(d_in, img, nin, hipMemcpyHostToDevice, ctx->stream));
  GH_TRY(gh_undistort_dev(p, d_in, channels, 1, nin, d_out, nout, fast));
  GH_HIP(ctx, hipMemcpyAsync(out, d_out, nout, hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GH_OK;
}
