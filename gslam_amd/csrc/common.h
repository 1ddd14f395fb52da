// Internal shared definitions for libgslam_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/gslam_hip.h"

struct gh_prof_pending {
  int slot;
  hipEvent_t start, stop;
};

struct gh_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string last_error;
  // profiling
  bool prof_on = false;
  std::vector<gh_prof_entry> prof_entries;
  std::vector<gh_prof_pending> prof_pending;
  std::vector<hipEvent_t> event_pool;
  // scratch owned by the context (grown on demand)
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  // pinned host staging for the small host-buffer entry points (one DMA each way instead of one per array)
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  // small pinned block of its own for what the solvers read back inside their loops (gh_readback_block): it never moves,
  // whatever gh_pinned is asked for meanwhile
  void* rb_pinned = nullptr;
  // grow-only arena reused by successive gh_ba_solve calls (local BA runs every keyframe: no malloc/free per call)
  void* ba_arena = nullptr;
  size_t ba_arena_bytes = 0;
  // grow-only arena of the graph solvers (gh_graph_solve / gh_pg_solve, graph_arena.h): one hipMalloc per context, not ~70 per solve
  void* pg_arena = nullptr;
  size_t pg_arena_bytes = 0;
  // every public entry point holds this for its whole call (GH_ENTER): a ctx shared by several Messenger worker threads
  // serialises on it; recursive because gh_ba_pnp calls gh_ba_solve
  std::recursive_mutex mu;
  int cu_count = 0;
  // linear solver of gh_ba_solve's reduced camera system (gh_ctx_set_ba_solver): 0 auto, 1 dense, 2 band (cyclic reduction)
  int ba_solver = 0;
  int ba_last_solver = 0;  // what the last gh_ba_solve / gh_ba_graph_solve used: 1 dense, 2 band (T tiles in ba_last_band_tiles)
  int ba_last_band_tiles = 0, ba_last_cam_span = 0;
  int ba_last_border_points = 0;  // long-range points kept out of the Schur complement as the arrowhead border
  int ba_last_border_cams = 0, ba_last_reordered = 0;  // cameras in the arrowhead border; 1 = the solver re-ordered the cameras (ba_order.hip)
  // band solver (chol_cr.hip): side stream for the work off its critical path, and the events that order the two
  // renumbered copy of a problem's camera-indexed arrays (ba.hip: ArrowProblem), kept between solves: a fresh 24 MB vector per
  // solve cost the upload path ~20 ms at C5 (first use of never-seen pageable memory by the copy engine's staging)
  std::vector<double> ba_order_pose;
  std::vector<int32_t> ba_order_dof, ba_order_ocam;
  hipStream_t cr_side = nullptr;
  std::vector<hipEvent_t> cr_events;
  // gh_ba_solve: marks the candidate cost's read-back (the host waits for it, not for the stream: see ba.hip)
};

// Entry guard of every public function that touches the device: serialises callers that share the context and makes the
// context's GPU the calling thread's current device (a ctx may be created on one thread and used on another, and a
// process may hold contexts for several GPUs; allocations, copies and launches below all go to the CURRENT device).
// The caller's current device is restored on exit: a host process that also drives another GPU through HIP / torch must
// not find its thread switched to ours after a gh_* call.
struct gh_device_guard {
  int prev = -1;
  explicit gh_device_guard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) (void)hipSetDevice(device);
    else prev = -1;  // nothing to restore
  }
  ~gh_device_guard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
struct gh_enter_guard {
  std::lock_guard<std::recursive_mutex> lock;
  gh_device_guard dev;
  explicit gh_enter_guard(gh_ctx* c) : lock(c->mu), dev(c->device) {}
};
#define GH_ENTER(ctx) gh_enter_guard _gh_enter_guard(ctx)

gh_status gh_set_error(gh_ctx* ctx, gh_status st, const char* fmt, ...);
gh_status gh_scratch(gh_ctx* ctx, size_t bytes, void** out);
gh_status gh_pinned(gh_ctx* ctx, size_t bytes, void** out);  // grow-only pinned host block owned by the context
gh_status gh_readback_block(gh_ctx* ctx, size_t bytes, void** out);  // <= 4096 bytes of pinned host memory that never move
int gh_prof_begin(gh_ctx* ctx, const char* name);  // returns pending index or -1
void gh_prof_end(gh_ctx* ctx, int pending);

#define GH_HIP(ctx, expr)                                                                  \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      return gh_set_error((ctx), GH_ERR_HIP, "%s failed: %s (%s:%d)", #expr,              \
                          hipGetErrorString(_e), __FILE__, __LINE__);                      \
  } while (0)

#define GH_CHECK_ARG(ctx, cond)                                                            \
  do {                                                                                     \
    if (!(cond))                                                                           \
      return gh_set_error((ctx), GH_ERR_ARG, "argument check failed: %s (%s:%d)", #cond,  \
                          __FILE__, __LINE__);                                             \
  } while (0)

#define GH_TRY(expr)                 \
  do {                               \
    gh_status _s = (expr);           \
    if (_s != GH_OK) return _s;      \
  } while (0)

// Launch a kernel on the ctx stream, timed under `name` when profiling is on.
#define GH_LAUNCH(ctx, name, kernel, grid, block, shmem, ...)                              \
  do {                                                                                     \
    int _p = gh_prof_begin((ctx), (name));                                                 \
    hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__);            \
    gh_prof_end((ctx), _p);                                                                \
    hipError_t _e = hipGetLastError();                                                     \
    if (_e != hipSuccess)                                                                  \
      return gh_set_error((ctx), GH_ERR_HIP, "launch %s failed: %s", (name),               \
                          hipGetErrorString(_e));                                          \
  } while (0)

static inline int gh_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }
