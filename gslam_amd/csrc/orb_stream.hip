// gh_orb_stream_*: the host-fed ORB extraction path behind the C ABI -- frames arrive in HOST memory chunk by chunk
// (a dataset reader / camera thread publishing FramePtrs: GSLAM/plugins/play/main.cpp:99-155), results are wanted in
// host memory (MapFrame::setKeyPoints, GSLAM/core/Map.h:309-321).  A `depth`-deep ring of slots, three HIP streams:
//
//   h2d      one flat DMA of the chunk (pinned staging of the slot, or the caller's own buffer) into the slot's HBM buffer
//   compute  [BGR(A) -> luma] + gh_orb_extract_dev of the chunk (ONE plan shared by all slots: this stream serialises them)
//   d2h      a pack kernel that writes, straight into the slot's pinned result block over PCIe, per-frame offsets and ONLY
//            the valid records of every frame back to back (exact-size "counts first, then count records" without a host
//            round trip in between)
//
// chained by events, so the copies of chunk i + 1 / i - 1 run under the kernels of chunk i.  Measured on the box
// (tools/pcie_probe.hip): the link gives 57 GB/s either way with one DMA stream, so a 1080p frame costs 36 us on the
// link against 20 us of kernels: the path is link-bound and the kernels hide under the copy.
#include <new>

#include "common.h"

namespace {

// frame f of the chunk: records [off[f], off[f + 1]) of the packed output.  grid = (n_frames, kPackSplit).
constexpr int kPackSplit = 4;
__global__ __launch_bounds__(256) void pack_results_kernel(const gh_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                          const int32_t* __restrict__ counts, int K, int n_frames,
                                                          int32_t* __restrict__ h_off, uint32_t* __restrict__ h_kps,
                                                          uint32_t* __restrict__ h_desc) {
  __shared__ int s_part[4];
  const int f = blockIdx.x, tid = threadIdx.x;
  int before = 0;
  for (int i = tid; i < f; i += 256) before += counts[i];
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
  if ((tid & 63) == 0) s_part[tid >> 6] = before;
  __syncthreads();
  before = s_part[0] + s_part[1] + s_part[2] + s_part[3];
  const int n = counts[f];
  if (blockIdx.y == 0 && tid == 0) {
    h_off[f] = before;
    if (f == n_frames - 1) h_off[n_frames] = before + n;
  }
  const uint32_t* sk = reinterpret_cast<const uint32_t*>(kps + (size_t)f * K);
  const uint32_t* sd = reinterpret_cast<const uint32_t*>(desc + (size_t)f * K * 32);
  uint32_t* dk = h_kps + (size_t)before * 7;
  uint32_t* dd = h_desc + (size_t)before * 8;
  const int nk = n * 7, nd = n * 8;
  for (int i = blockIdx.y * 256 + tid; i < nk; i += 256 * kPackSplit) dk[i] = sk[i];
  for (int i = blockIdx.y * 256 + tid; i < nd; i += 256 * kPackSplit) dd[i] = sd[i];
}

struct Slot {
  uint8_t* h_in = nullptr;   // pinned staging: chunk x frame_stride
  uint8_t* d_in = nullptr;   // the same in HBM (+ pad)
  uint8_t* d_gray = nullptr; // colour input only: luma frames
  gh_keypoint* d_kps = nullptr;
  uint8_t* d_desc = nullptr;
  int32_t* d_counts = nullptr;
  uint8_t* h_out = nullptr;  // pinned, mapped: [offsets][keypoints][descriptors]
  uint8_t* h_out_dev = nullptr;
  hipEvent_t ev_begin = nullptr, ev_h2d = nullptr, ev_extract = nullptr, ev_done = nullptr;
  int64_t ticket = -1;
  int n_frames = 0;
};

}  // namespace

struct gh_orb_stream {
  gh_ctx* ctx = nullptr;    // the caller's context: error text, device
  gh_ctx* cctx = nullptr;   // private context whose stream is the compute stream (the plan launches on it)
  gh_orb_plan* plan = nullptr;
  hipStream_t s_h2d = nullptr, s_d2h = nullptr;
  int w = 0, h = 0, ch = 1, row_stride = 0, chunk = 0, depth = 0, K = 0, gray_pitch = 0;
  size_t frame_stride = 0, off_kps = 0, off_desc = 0, out_bytes = 0;
  std::vector<Slot> slots;
  int64_t next_ticket = 0;
  std::mutex mu;
};

static void stream_free(gh_orb_stream* s) {
  if (!s) return;
  if (s->ctx) (void)hipSetDevice(s->ctx->device);
  if (s->cctx) hipStreamSynchronize(s->cctx->stream);
  if (s->s_h2d) hipStreamSynchronize(s->s_h2d);
  if (s->s_d2h) hipStreamSynchronize(s->s_d2h);
  if (s->plan) gh_orb_plan_destroy(s->plan);
  for (Slot& sl : s->slots) {
    if (sl.h_in) hipHostFree(sl.h_in);
    if (sl.h_out) hipHostFree(sl.h_out);
    void* dev[] = {sl.d_in, sl.d_gray, sl.d_kps, sl.d_desc, sl.d_counts};
    for (void* p : dev)
      if (p) hipFree(p);
    hipEvent_t ev[] = {sl.ev_begin, sl.ev_h2d, sl.ev_extract, sl.ev_done};
    for (hipEvent_t e : ev)
      if (e) hipEventDestroy(e);
  }
  if (s->s_h2d) hipStreamDestroy(s->s_h2d);
  if (s->s_d2h) hipStreamDestroy(s->s_d2h);
  if (s->cctx) gh_ctx_destroy(s->cctx);
  delete s;
}

extern "C" void gh_orb_stream_destroy(gh_orb_stream* s) { stream_free(s); }

extern "C" gh_status gh_orb_stream_create(gh_ctx* ctx, int width, int height, int channels, int row_stride,
                                          size_t frame_stride, int chunk_frames, int depth, const gh_orb_params* params,
                                          gh_orb_stream** out) {
  if (!ctx || !out) return GH_ERR_ARG;
  GH_ENTER(ctx);
  *out = nullptr;
  GH_CHECK_ARG(ctx, channels == 1 || channels == 3 || channels == 4);
  GH_CHECK_ARG(ctx, width > 0 && height > 0 && row_stride >= width * channels && frame_stride >= (size_t)row_stride * height);
  GH_CHECK_ARG(ctx, chunk_frames >= 1 && chunk_frames <= 65535 && depth >= 1 && depth <= 16);
  gh_orb_stream* s = new (std::nothrow) gh_orb_stream();
  if (!s) return GH_ERR_NOMEM;
  s->ctx = ctx;
  s->w = width;
  s->h = height;
  s->ch = channels;
  s->row_stride = row_stride;
  s->frame_stride = frame_stride;
  s->chunk = chunk_frames;
  s->depth = depth;
  gh_orb_params prm;
  gh_orb_default_params(&prm);
  if (params) prm = *params;
  s->K = prm.n_features;
  s->gray_pitch = (width + 63) & ~63;
  gh_status st = gh_ctx_create(ctx->device, &s->cctx);
  if (st == GH_OK) st = gh_orb_plan_create(s->cctx, width, height, chunk_frames, &prm, &s->plan);
  if (st != GH_OK) {
    gh_set_error(ctx, st, "gh_orb_stream_create: %s", s->cctx ? gh_last_error(s->cctx) : "no private context");
    stream_free(s);
    return st;
  }
  const size_t K = (size_t)s->K, C = (size_t)chunk_frames;
  s->off_kps = ((C + 1) * sizeof(int32_t) + 255) & ~(size_t)255;
  s->off_desc = s->off_kps + ((C * K * sizeof(gh_keypoint) + 255) & ~(size_t)255);
  s->out_bytes = s->off_desc + C * K * 32;
  s->slots.resize((size_t)depth);
  // The two copy streams are HIGH priority: the runtime multiplexes the streams of a process onto a few hardware queues
  // (4 by default) PER PRIORITY LEVEL, and two streams that land on one queue execute in submission order -- in a process
  // that already owns several streams (measured inside bench.py: torch + the other legs) the upload of chunk i + 1 then
  // queued behind the kernels of chunk i and the pipeline ran at copy + compute (32 GB/s) instead of max(copy, compute)
  // (52 GB/s).  A priority of their own gives the copies queues of their own, and it is also the right order: a late
  // copy stalls the whole ring, a late kernel does not.
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  bool ok = hipStreamCreateWithPriority(&s->s_h2d, hipStreamNonBlocking, prio_greatest) == hipSuccess &&
            hipStreamCreateWithPriority(&s->s_d2h, hipStreamNonBlocking, prio_greatest) == hipSuccess;
  for (Slot& sl : s->slots) {
    if (!ok) break;
    ok = hipHostMalloc((void**)&sl.h_in, C * frame_stride, hipHostMallocDefault) == hipSuccess &&
         hipHostMalloc((void**)&sl.h_out, s->out_bytes, hipHostMallocDefault) == hipSuccess &&
         hipHostGetDevicePointer((void**)&sl.h_out_dev, sl.h_out, 0) == hipSuccess &&
         hipMalloc((void**)&sl.d_in, C * frame_stride + 256) == hipSuccess &&
         (channels == 1 || hipMalloc((void**)&sl.d_gray, C * (size_t)s->gray_pitch * height + 256) == hipSuccess) &&
         hipMalloc((void**)&sl.d_kps, C * K * sizeof(gh_keypoint)) == hipSuccess &&
         hipMalloc((void**)&sl.d_desc, C * K * 32) == hipSuccess && hipMalloc((void**)&sl.d_counts, C * sizeof(int32_t)) == hipSuccess &&
         hipEventCreate(&sl.ev_begin) == hipSuccess && hipEventCreateWithFlags(&sl.ev_h2d, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&sl.ev_extract, hipEventDisableTiming) == hipSuccess && hipEventCreate(&sl.ev_done) == hipSuccess;
    if (ok) memset(sl.h_out, 0, s->off_kps);
  }
  if (!ok) {
    gh_set_error(ctx, GH_ERR_NOMEM, "gh_orb_stream_create: allocating %d slots of %d frames failed: %s", depth, chunk_frames,
                 hipGetErrorString(hipGetLastError()));
    stream_free(s);
    return GH_ERR_NOMEM;
  }
  *out = s;
  return GH_OK;
}

extern "C" gh_status gh_orb_stream_staging(gh_orb_stream* s, uint8_t** host_pinned) {
  if (!s || !host_pinned) return GH_ERR_ARG;
  gh_ctx* ctx = s->ctx;
  GH_ENTER(ctx);
  std::lock_guard<std::mutex> lock(s->mu);
  Slot& sl = s->slots[(size_t)(s->next_ticket % s->depth)];
  if (sl.ticket >= 0) GH_HIP(ctx, hipEventSynchronize(sl.ev_h2d));  // the previous upload out of this block is over
  *host_pinned = sl.h_in;
  return GH_OK;
}

extern "C" gh_status gh_orb_stream_submit(gh_orb_stream* s, const uint8_t* frames_host, int n_frames, int64_t* ticket) {
  if (!s) return GH_ERR_ARG;
  gh_ctx* ctx = s->ctx;
  GH_ENTER(ctx);
  std::lock_guard<std::mutex> lock(s->mu);
  GH_CHECK_ARG(ctx, n_frames >= 1 && n_frames <= s->chunk);
  Slot& sl = s->slots[(size_t)(s->next_ticket % s->depth)];
  // ring reuse: everything the previous ticket of this slot put on the streams must be over (its results are then
  // overwritten: collect ticket t before submitting ticket t + depth)
  if (sl.ticket >= 0) GH_HIP(ctx, hipEventSynchronize(sl.ev_done));
  const uint8_t* src = frames_host ? frames_host : sl.h_in;
  // the caller's last frame may end at its last pixel
  const size_t bytes = (size_t)(n_frames - 1) * s->frame_stride + (size_t)(s->h - 1) * s->row_stride + (size_t)s->w * s->ch;
  GH_HIP(ctx, hipEventRecord(sl.ev_begin, s->s_h2d));
  GH_HIP(ctx, hipMemcpyAsync(sl.d_in, src, bytes, hipMemcpyHostToDevice, s->s_h2d));
  GH_HIP(ctx, hipEventRecord(sl.ev_h2d, s->s_h2d));
  hipStream_t sc = s->cctx->stream;
  GH_HIP(ctx, hipStreamWaitEvent(sc, sl.ev_h2d, 0));
  gh_status st = GH_OK;
  if (s->ch == 1) {
    st = gh_orb_extract_dev(s->plan, sl.d_in, n_frames, s->frame_stride, s->row_stride, sl.d_kps, sl.d_desc, sl.d_counts);
  } else {
    st = gh_bgr_to_gray_batch_dev(s->cctx, sl.d_in, s->w, s->h, s->ch, s->row_stride, s->frame_stride, n_frames, sl.d_gray,
                                  s->gray_pitch, (size_t)s->gray_pitch * s->h);
    if (st == GH_OK)
      st = gh_orb_extract_dev(s->plan, sl.d_gray, n_frames, (size_t)s->gray_pitch * s->h, s->gray_pitch, sl.d_kps, sl.d_desc,
                              sl.d_counts);
  }
  if (st != GH_OK) return gh_set_error(ctx, st, "gh_orb_stream_submit: %s", gh_last_error(s->cctx));
  GH_HIP(ctx, hipEventRecord(sl.ev_extract, sc));
  GH_HIP(ctx, hipStreamWaitEvent(s->s_d2h, sl.ev_extract, 0));
  hipLaunchKernelGGL(pack_results_kernel, dim3(n_frames, kPackSplit), dim3(256), 0, s->s_d2h, sl.d_kps, sl.d_desc, sl.d_counts,
                     s->K, n_frames, reinterpret_cast<int32_t*>(sl.h_out_dev), reinterpret_cast<uint32_t*>(sl.h_out_dev + s->off_kps),
                     reinterpret_cast<uint32_t*>(sl.h_out_dev + s->off_desc));
  GH_HIP(ctx, hipGetLastError());
  GH_HIP(ctx, hipEventRecord(sl.ev_done, s->s_d2h));
  // the next user of the plan's workspace (the following chunk) is ordered behind this extraction by the compute
  // stream itself; the next user of THIS slot's buffers waits for ev_done above
  sl.ticket = s->next_ticket++;
  sl.n_frames = n_frames;
  if (ticket) *ticket = sl.ticket;
  return GH_OK;
}

static Slot* find_ticket(gh_orb_stream* s, int64_t ticket) {
  if (ticket < 0 || ticket >= s->next_ticket || ticket < s->next_ticket - s->depth) return nullptr;
  Slot& sl = s->slots[(size_t)(ticket % s->depth)];
  return sl.ticket == ticket ? &sl : nullptr;
}

// poll / collect do NOT hold the context's lock while they wait: a producer thread keeps submitting on the same context
// while a consumer thread blocks in collect.  The error text is written under the lock.
static gh_status stream_fail(gh_orb_stream* s, gh_status st, const char* what, long long ticket, hipError_t e) {
  std::lock_guard<std::recursive_mutex> lock(s->ctx->mu);
  if (e != hipSuccess) return gh_set_error(s->ctx, st, "%s: ticket %lld: %s", what, ticket, hipGetErrorString(e));
  return gh_set_error(s->ctx, st, "%s: ticket %lld is not in flight (ring depth %d, next ticket %lld)", what, ticket, s->depth,
                      (long long)s->next_ticket);
}

extern "C" gh_status gh_orb_stream_poll(gh_orb_stream* s, int64_t ticket, int* ready) {
  if (!s || !ready) return GH_ERR_ARG;
  gh_device_guard dev(s->ctx->device);
  Slot* sl;
  {
    std::lock_guard<std::mutex> lock(s->mu);
    sl = find_ticket(s, ticket);
  }
  if (!sl) return stream_fail(s, GH_ERR_ARG, "gh_orb_stream_poll", ticket, hipSuccess);
  const hipError_t e = hipEventQuery(sl->ev_done);
  if (e != hipSuccess && e != hipErrorNotReady) return stream_fail(s, GH_ERR_HIP, "gh_orb_stream_poll", ticket, e);
  *ready = e == hipSuccess ? 1 : 0;
  return GH_OK;
}

extern "C" gh_status gh_orb_stream_collect(gh_orb_stream* s, int64_t ticket, gh_orb_stream_result* out) {
  if (!s || !out) return GH_ERR_ARG;
  gh_device_guard dev(s->ctx->device);
  Slot* sl;
  {
    std::lock_guard<std::mutex> lock(s->mu);
    sl = find_ticket(s, ticket);
  }
  if (!sl) return stream_fail(s, GH_ERR_ARG, "gh_orb_stream_collect", ticket, hipSuccess);
  const hipError_t e = hipEventSynchronize(sl->ev_done);
  if (e != hipSuccess) return stream_fail(s, GH_ERR_HIP, "gh_orb_stream_collect", ticket, e);
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, sl->ev_begin, sl->ev_done) != hipSuccess) ms = -1.f;
  out->n_frames = sl->n_frames;
  out->offsets = reinterpret_cast<const int32_t*>(sl->h_out);
  out->kps = reinterpret_cast<const gh_keypoint*>(sl->h_out + s->off_kps);
  out->desc = sl->h_out + s->off_desc;
  out->gpu_ms = ms;
  return GH_OK;
}

// The extraction plan behind the stream, for the per-plan settings (gh_orb_plan_set_pattern / gh_orb_plan_set_steering): call
// them while no ticket is outstanding.  The stream owns the plan.
extern "C" gh_orb_plan* gh_orb_stream_plan(gh_orb_stream* stream) { return stream ? stream->plan : nullptr; }
