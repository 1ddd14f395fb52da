// Context, device memory helpers and live HIP-event profiling for libgslam_hip.so.
#include <stdarg.h>

#include "common.h"

gh_status gh_set_error(gh_ctx* ctx, gh_status st, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->last_error = buf;
  return st;
}

// 2 (round 5): gh_graph_problem grew intrinsics / intrinsics_free (round 4), gh_ctx_last_ba_solver reports GH_BA_SOLVER_ARROW,
//               gh_arrow_solve_dev, gh_bf_match*_bytes; a host built against version 1 must be rebuilt
extern "C" int gh_abi_version(void) { return 2; }

extern "C" gh_status gh_ctx_create(int device, gh_ctx** out) {
  if (!out) return GH_ERR_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return GH_ERR_HIP;
  if (hipSetDevice(device) != hipSuccess) return GH_ERR_HIP;
  gh_ctx* c = new (std::nothrow) gh_ctx();
  if (!c) return GH_ERR_NOMEM;
  c->device = device;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return GH_ERR_HIP;
  }
  c->stream = c->own_stream;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->cu_count = prop.multiProcessorCount;
  *out = c;
  return GH_OK;
}

extern "C" void gh_ctx_destroy(gh_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  for (auto& p : ctx->prof_pending) {
    hipEventDestroy(p.start);
    hipEventDestroy(p.stop);
  }
  for (auto e : ctx->event_pool) hipEventDestroy(e);
  if (ctx->scratch) hipFree(ctx->scratch);
  if (ctx->pinned) hipHostFree(ctx->pinned);
  if (ctx->rb_pinned) hipHostFree(ctx->rb_pinned);
  if (ctx->ba_arena) hipFree(ctx->ba_arena);
  if (ctx->pg_arena) hipFree(ctx->pg_arena);
  for (auto e : ctx->cr_events) hipEventDestroy(e);
  if (ctx->cr_side) hipStreamDestroy(ctx->cr_side);
  if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

extern "C" const char* gh_last_error(const gh_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

extern "C" gh_status gh_ctx_set_ba_solver(gh_ctx* ctx, int solver) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, solver >= 0 && solver <= 2);
  ctx->ba_solver = solver;
  return GH_OK;
}

extern "C" int gh_ctx_last_ba_solver(gh_ctx* ctx, int* band_tiles, int* cam_span) {
  if (!ctx) return 0;
  GH_ENTER(ctx);
  if (band_tiles) *band_tiles = ctx->ba_last_band_tiles;
  if (cam_span) *cam_span = ctx->ba_last_cam_span;
  return ctx->ba_last_solver;
}

extern "C" int gh_ctx_last_ba_order(gh_ctx* ctx, int* border_cams, int* reordered) {
  if (!ctx) return 0;
  GH_ENTER(ctx);
  if (border_cams) *border_cams = ctx->ba_last_border_cams;
  if (reordered) *reordered = ctx->ba_last_reordered;
  return ctx->ba_last_solver != 0;
}

extern "C" int gh_ctx_last_ba_border_points(gh_ctx* ctx) {
  if (!ctx) return 0;
  GH_ENTER(ctx);
  return ctx->ba_last_border_points;
}

extern "C" gh_status gh_ctx_set_stream(gh_ctx* ctx, void* hip_stream) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  ctx->stream = (hipStream_t)hip_stream;
  return GH_OK;
}

extern "C" gh_status gh_ctx_use_own_stream(gh_ctx* ctx) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  ctx->stream = ctx->own_stream;
  return GH_OK;
}

extern "C" void* gh_ctx_stream(gh_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" gh_status gh_ctx_sync(gh_ctx* ctx) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GH_OK;
}

extern "C" gh_status gh_device_info(gh_ctx* ctx, int* cu_count, int* clock_khz, size_t* hbm_bytes, char* name,
                                    int name_cap) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  hipDeviceProp_t prop;
  GH_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (clock_khz) *clock_khz = prop.clockRate;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  if (name && name_cap > 0) {
    snprintf(name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
  }
  return GH_OK;
}

extern "C" gh_status gh_dev_alloc(gh_ctx* ctx, size_t bytes, void** out_dev) {
  if (!ctx || !out_dev) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_HIP(ctx, hipSetDevice(ctx->device));
  hipError_t e = hipMalloc(out_dev, bytes ? bytes : 1);
  if (e != hipSuccess) return gh_set_error(ctx, GH_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
  return GH_OK;
}

extern "C" gh_status gh_dev_free(gh_ctx* ctx, void* dev) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  if (dev) GH_HIP(ctx, hipFree(dev));
  return GH_OK;
}

extern "C" gh_status gh_dev_upload(gh_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  if (bytes == 0) return GH_OK;
  GH_CHECK_ARG(ctx, dst_dev && src_host);
  GH_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GH_OK;
}

extern "C" gh_status gh_dev_download(gh_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  if (bytes == 0) return GH_OK;
  GH_CHECK_ARG(ctx, dst_host && src_dev);
  GH_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GH_OK;
}

extern "C" gh_status gh_dev_memset(gh_ctx* ctx, void* dst_dev, int value, size_t bytes) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  if (bytes == 0) return GH_OK;
  GH_HIP(ctx, hipMemsetAsync(dst_dev, value, bytes, ctx->stream));
  return GH_OK;
}

extern "C" gh_status gh_host_alloc_pinned(gh_ctx* ctx, size_t bytes, void** out_host) {
  if (!ctx || !out_host) return GH_ERR_ARG;
  GH_ENTER(ctx);
  hipError_t e = hipHostMalloc(out_host, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) return gh_set_error(ctx, GH_ERR_NOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
  return GH_OK;
}

extern "C" gh_status gh_host_free_pinned(gh_ctx* ctx, void* host) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  if (host) GH_HIP(ctx, hipHostFree(host));
  return GH_OK;
}

gh_status gh_scratch(gh_ctx* ctx, size_t bytes, void** out) {
  if (bytes > ctx->scratch_bytes) {
    // The stream may still be using the old block: drain before replacing it.
    GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->scratch) GH_HIP(ctx, hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipMalloc(&ctx->scratch, want);
    if (e != hipSuccess) return gh_set_error(ctx, GH_ERR_NOMEM, "scratch hipMalloc(%zu): %s", want, hipGetErrorString(e));
    ctx->scratch_bytes = want;
  }
  *out = ctx->scratch;
  return GH_OK;
}

gh_status gh_readback_block(gh_ctx* ctx, size_t bytes, void** out) {
  if (bytes > 4096) return gh_set_error(ctx, GH_ERR_ARG, "gh_readback_block: %zu bytes", bytes);
  if (!ctx->rb_pinned) {
    // coherent (fine-grained) by request: gh_ba_solve polls this block while the kernel that writes it is still in the stream
    hipError_t e = hipHostMalloc(&ctx->rb_pinned, 4096, hipHostMallocCoherent);
    if (e != hipSuccess) {
      ctx->rb_pinned = nullptr;
      return gh_set_error(ctx, GH_ERR_NOMEM, "hipHostMalloc(4096): %s", hipGetErrorString(e));
    }
  }
  *out = ctx->rb_pinned;
  return GH_OK;
}

// Give back what the context has grown for past calls: scratch, pinned staging and the two solver arenas (a single large
// loop closure or graph would otherwise pin its high-water mark -- up to 64 MB of host memory and tens of MB of HBM -- for
// the life of the context).  Everything is re-grown on demand; device-resident graphs (gh_ba_graph) own their memory and
// are not touched.
extern "C" gh_status gh_ctx_trim(gh_ctx* ctx) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->cr_side) GH_HIP(ctx, hipStreamSynchronize(ctx->cr_side));
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  ctx->scratch = nullptr;
  ctx->scratch_bytes = 0;
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  ctx->pinned = nullptr;
  ctx->pinned_bytes = 0;
  std::vector<double>().swap(ctx->ba_order_pose);
  std::vector<int32_t>().swap(ctx->ba_order_dof);
  std::vector<int32_t>().swap(ctx->ba_order_ocam);
  if (ctx->ba_arena) (void)hipFree(ctx->ba_arena);
  ctx->ba_arena = nullptr;
  ctx->ba_arena_bytes = 0;
  if (ctx->pg_arena) (void)hipFree(ctx->pg_arena);
  ctx->pg_arena = nullptr;
  ctx->pg_arena_bytes = 0;
  return GH_OK;
}

gh_status gh_pinned(gh_ctx* ctx, size_t bytes, void** out) {
  if (bytes > ctx->pinned_bytes) {
    GH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // a copy may still be reading the old block
    if (ctx->pinned) GH_HIP(ctx, hipHostFree(ctx->pinned));
    ctx->pinned = nullptr;
    ctx->pinned_bytes = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault);
    if (e != hipSuccess) return gh_set_error(ctx, GH_ERR_NOMEM, "hipHostMalloc(%zu): %s", want, hipGetErrorString(e));
    ctx->pinned_bytes = want;
  }
  *out = ctx->pinned;
  return GH_OK;
}

// ------------------------------------------------------------------ profiling ----------
static hipEvent_t take_event(gh_ctx* ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

static void resolve_pending(gh_ctx* ctx) {
  if (ctx->prof_pending.empty()) return;
  hipStreamSynchronize(ctx->stream);
  for (auto& p : ctx->prof_pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
      ctx->prof_entries[p.slot].launches += 1;
      ctx->prof_entries[p.slot].total_ms += ms;
    }
    ctx->event_pool.push_back(p.start);
    ctx->event_pool.push_back(p.stop);
  }
  ctx->prof_pending.clear();
}

int gh_prof_begin(gh_ctx* ctx, const char* name) {
  if (!ctx->prof_on) return -1;
  int slot = -1;
  for (size_t i = 0; i < ctx->prof_entries.size(); ++i)
    if (strncmp(ctx->prof_entries[i].name, name, sizeof(ctx->prof_entries[i].name)) == 0) {
      slot = (int)i;
      break;
    }
  if (slot < 0) {
    gh_prof_entry e;
    memset(&e, 0, sizeof(e));
    strncpy(e.name, name, sizeof(e.name) - 1);
    ctx->prof_entries.push_back(e);
    slot = (int)ctx->prof_entries.size() - 1;
  }
  if (ctx->prof_pending.size() >= 4096) resolve_pending(ctx);
  gh_prof_pending p;
  p.slot = slot;
  p.start = take_event(ctx);
  p.stop = take_event(ctx);
  hipEventRecord(p.start, ctx->stream);
  ctx->prof_pending.push_back(p);
  return (int)ctx->prof_pending.size() - 1;
}

void gh_prof_end(gh_ctx* ctx, int pending) {
  if (pending < 0) return;
  hipEventRecord(ctx->prof_pending[pending].stop, ctx->stream);
}

extern "C" gh_status gh_prof_enable(gh_ctx* ctx, int on) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  resolve_pending(ctx);
  if (on) ctx->prof_entries.clear();
  ctx->prof_on = on != 0;
  return GH_OK;
}

extern "C" gh_status gh_prof_collect(gh_ctx* ctx, gh_prof_entry* out, int cap, int* n) {
  if (!ctx || !n) return GH_ERR_ARG;
  GH_ENTER(ctx);
  resolve_pending(ctx);
  int m = (int)ctx->prof_entries.size();
  if (m > cap) m = cap;
  for (int i = 0; i < m; ++i) out[i] = ctx->prof_entries[i];
  *n = m;
  return GH_OK;
}
