// Multi-GPU exchange step of the front end behind the C ABI: frames are sharded over the GPUs of one
// node, one process per GPU; after extraction every rank needs every frame's records, and after matching every rank
// may want every match row.  Both are all-gathers of fixed-size per-frame records -- pure data movement, so the
// gathered buffers are bit-identical on every rank and for every world size.
//
// The reference has no transport at all (its Messenger is in-process: GSLAM/core/Messenger.h:455-468,687-715); this is
// new work that a C++ host reaches through gh_comm_* with no Python and no torch in the process.
//
// Two transports behind one interface:
//   RCCL   ncclAllGather on the communicator's own stream (xGMI).  librccl.so.1 is resolved with dlopen when the first
//          communicator is created, so single-GPU users and hosts without RCCL never load it; a process that already
//          holds RCCL (torch.distributed) shares that copy (same SONAME).
//   IPC    same-node direct writes: every rank maps every peer's gathered buffer through HIP IPC and pushes its own
//          slice into all of them with device-to-device copies (on a full xGMI mesh that drives all links at once);
//          rendezvous, handle exchange and the barriers go through a POSIX shared-memory segment.  Works when several
//          ranks share one GPU (RCCL refuses that: "Duplicate GPU detected"), which is how the 2-process test on a
//          1-GPU box runs the real kernels.  Asynchronous like the RCCL path: no host thread waits for a GPU or for a
//          peer during an exchange.  The ordering between the ranks' streams goes through two rows of flags in the
//          rendezvous segment (host memory, registered with HIP in every process, so every GPU reads and writes it
//          coherently):
//              ready[r] = k   written by a one-thread kernel on r's CONTEXT stream when exchange k is issued: everything
//                             r enqueued before has run -- its send data exists and its gathered buffers are no longer
//                             being read, so peers may overwrite them;
//              done[r]  = k   written on r's COMMUNICATOR stream after its pushes of exchange k: r's slice has landed in
//                             every peer.
//          r's communicator stream starts with a kernel that polls ready[*] >= k, then pushes; gh_comm_wait enqueues a
//          kernel on the context stream that polls done[*] >= k.  Every poll is bounded (GSLAM_HIP_COMM_TIMEOUT_S):
//          a rank that gives up sets `failed`, which every other poll and the next gh_comm_* call on any rank sees
//          (gh_comm_status).  GSLAM_HIP_IPC_SYNC=1 selects the older protocol (host barriers around the pushes), kept
//          for diagnosis.
//
// Stream model (both transports): a gather issued by gh_allgather* is ordered AFTER everything already enqueued on the
// context's stream, runs on the communicator's stream, and gh_comm_wait orders the context's stream after it -- so
// kernels enqueued between the two calls overlap with the transfer and the host never blocks.
#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <new>

#include "common.h"

namespace {

// ------------------------------------------------------------------ RCCL entry points (dlopen) ----------
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (api.handle) break;
    }
    if (!api.handle) return;
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
    api.AllGather = (decltype(api.AllGather))dlsym(api.handle, "ncclAllGather");
    api.GroupStart = (decltype(api.GroupStart))dlsym(api.handle, "ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.handle, "ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.GroupStart && api.GroupEnd &&
             api.GetErrorString;
  });
  return api;
}

// ------------------------------------------------------------------ IPC rendezvous segment ----------
constexpr int kMaxWorld = 64;
constexpr int kMaxBuffers = 16;

struct alignas(64) FlagLine {
  uint32_t v;
  uint32_t pad[15];
};

struct ShmSegment {
  std::atomic<uint32_t> magic;
  std::atomic<int> arrived;          // sense-reversing barrier
  std::atomic<int> generation;
  std::atomic<int> attached;
  std::atomic<int> failed;           // any rank (host or a polling kernel) that gives up sets this so the others stop waiting
  hipIpcMemHandle_t handles[kMaxWorld];
  uint64_t sizes[kMaxWorld];
  FlagLine ready[kMaxWorld];         // exchange number up to which rank r's buffers may be overwritten / its send data exists
  FlagLine done[kMaxWorld];          // exchange number up to which rank r's pushes have landed everywhere
};
static_assert(sizeof(std::atomic<int>) == sizeof(int), "the polling kernels read `failed` as a plain int");

// one thread: everything enqueued before on this stream is complete -> publish exchange number k
__global__ void comm_flag_set_kernel(uint32_t* flag, uint32_t k) {
  __atomic_thread_fence(__ATOMIC_RELEASE);  // (system scope: the default of the builtin)
  __hip_atomic_store(flag, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// lane r < world polls flags[r] >= k (wrap-safe); bounded by `timeout_ticks` of the 100 MHz wall clock and by `failed`
__global__ void comm_flag_wait_kernel(const FlagLine* flags, int world, uint32_t k, unsigned long long timeout_ticks, int* failed) {
  const int r = threadIdx.x;
  if (r < world) {
    const unsigned long long t0 = wall_clock64();
    unsigned spins = 0;
    while ((int32_t)(__hip_atomic_load(&flags[r].v, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - k) < 0) {
      if ((++spins & 63u) == 0) {
        if (__hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
        if (wall_clock64() - t0 > timeout_ticks) {
          __hip_atomic_store(failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
      }
      __builtin_amdgcn_s_sleep(64);
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

struct IpcBuffer {
  void* mine = nullptr;                 // world * bytes_per_rank, hipMalloc'ed here
  void* peer[kMaxWorld] = {nullptr};    // peer[r] == r's `mine` mapped into this process (peer[rank] == mine)
  size_t bytes_per_rank = 0;
};

}  // namespace

struct gh_comm {
  gh_ctx* ctx = nullptr;
  int rank = 0, world = 1;
  int transport = 0;  // 0 RCCL, 1 IPC
  hipStream_t stream = nullptr;
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  bool pending = false;
  // RCCL
  ncclComm_t nccl = nullptr;
  std::vector<void*> plain_buffers;
  // IPC
  ShmSegment* shm = nullptr;
  std::string shm_name;
  int local_generation = 0;
  std::vector<IpcBuffer> ipc_buffers;
  double timeout_s = 60.0;
  bool ipc_sync = false;           // GSLAM_HIP_IPC_SYNC=1: host barriers instead of the flag kernels
  ShmSegment* shm_dev = nullptr;   // the segment as the GPU sees it (hipHostRegister)
  bool shm_registered = false;
  uint32_t round = 0;              // number of the exchange in flight / last issued
  // One stream per peer for the pushes when there is more than one peer (GSLAM_HIP_IPC_PEER_STREAMS=0 / 1 forces it): copies on ONE
  // stream run one at a time, i.e. over one xGMI link at a time; the mesh has a link to every peer (7 x ~153 GB/s), so the
  // slices to the 7 peers go out together -- the direct peer all-gather SURVEY 8(e) prefers over a ring.
  bool peer_streams = true;
  hipStream_t pstream[kMaxWorld] = {nullptr};
  hipEvent_t pev[kMaxWorld] = {nullptr};
  hipEvent_t ev_fork = nullptr;
};

namespace {

gh_status rccl_fail(gh_ctx* ctx, const char* what, ncclResult_t r) {
  return gh_set_error(ctx, GH_ERR_HIP, "%s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(r) : "?");
}

#define GH_NCCL(ctx, expr)                                 \
  do {                                                     \
    ncclResult_t _r = (expr);                              \
    if (_r != ncclSuccess) return rccl_fail((ctx), #expr, _r); \
  } while (0)

// Host barrier over the ranks attached to the segment; false on timeout or when a peer has failed.
bool shm_barrier(gh_comm* c) {
  ShmSegment* s = c->shm;
  const int gen = c->local_generation;
  if (s->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
    s->arrived.store(0, std::memory_order_relaxed);
    s->generation.store(gen + 1, std::memory_order_release);
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (s->generation.load(std::memory_order_acquire) == gen) {
      if (s->failed.load(std::memory_order_relaxed)) return false;
      if (++spins > 2000) {
        sched_yield();
        if ((spins & 1023) == 0 &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) {
          s->failed.store(1, std::memory_order_relaxed);
          return false;
        }
      }
    }
  }
  c->local_generation = gen + 1;
  return true;
}

gh_status comm_common_init(gh_ctx* ctx, gh_comm* c) {
  GH_HIP(ctx, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  GH_HIP(ctx, hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
  GH_HIP(ctx, hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
  return GH_OK;
}

constexpr uint32_t kShmMagic = 0x47534C4Du;

ShmSegment* map_segment(int fd) {
  void* m = mmap(nullptr, sizeof(ShmSegment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  return m == MAP_FAILED ? nullptr : (ShmSegment*)m;
}

// Rendezvous on the named segment.  The name is chosen by the launcher and may be the name of a segment a crashed run
// left behind (bench.py derives it from MASTER_PORT), and ranks start in any order, so a rank >= 1 can find the OLD
// segment before rank 0 has replaced it.  Rules that make this safe:
//   rank 0   poisons whatever segment exists under the name (magic = 0, failed = 1: ranks parked in it leave), unlinks
//            it, creates a fresh one (O_EXCL) and publishes the magic last;
//   rank r   attaches only to a segment that looks unused (magic set, generation 0, nobody failed, fewer than `world`
//            ranks attached); whenever the segment it sits in turns out to be dead -- poisoned, or replaced under the
//            name (different inode) -- it lets go and opens the name again, until the overall timeout.
gh_status ipc_rendezvous(gh_ctx* ctx, gh_comm* c) {
  const char* name = c->shm_name.c_str();
  const auto t0 = std::chrono::steady_clock::now();
  auto expired = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s; };
  if (c->rank == 0) {
    int old = shm_open(name, O_RDWR, 0600);
    if (old >= 0) {
      struct stat sb;
      if (fstat(old, &sb) == 0 && (size_t)sb.st_size >= sizeof(ShmSegment)) {
        if (ShmSegment* stale = map_segment(old)) {
          stale->magic.store(0, std::memory_order_release);
          stale->failed.store(1, std::memory_order_release);
          munmap(stale, sizeof(ShmSegment));
        }
      }
      close(old);
    }
    shm_unlink(name);
    int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(ShmSegment)) != 0) {
      if (fd >= 0) close(fd);
      return gh_set_error(ctx, GH_ERR_HIP, "IPC transport: cannot create rendezvous segment %s", name);
    }
    c->shm = map_segment(fd);
    close(fd);
    if (!c->shm) return gh_set_error(ctx, GH_ERR_HIP, "IPC transport: mmap of the rendezvous segment failed");
    c->shm->arrived.store(0);
    c->shm->generation.store(0);
    c->shm->attached.store(1);
    c->shm->failed.store(0);
    for (int r = 0; r < kMaxWorld; ++r) c->shm->ready[r].v = c->shm->done[r].v = 0;
    c->shm->magic.store(kShmMagic, std::memory_order_release);
    c->local_generation = 0;
    if (!shm_barrier(c)) return gh_set_error(ctx, GH_ERR_HIP, "IPC transport: rendezvous timed out (%d ranks expected)", c->world);
    return GH_OK;
  }
  while (!expired()) {
    int fd = shm_open(name, O_RDWR, 0600);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0 || (size_t)sb.st_size < sizeof(ShmSegment)) {  // not created / not sized yet
      if (fd >= 0) close(fd);
      usleep(1000);
      continue;
    }
    const ino_t ino = sb.st_ino;
    ShmSegment* seg = map_segment(fd);
    close(fd);
    if (!seg) return gh_set_error(ctx, GH_ERR_HIP, "IPC transport: mmap of the rendezvous segment failed");
    // still the segment the name points to?
    auto replaced = [&] {
      int f2 = shm_open(name, O_RDWR, 0600);
      if (f2 < 0) return true;  // unlinked: rank 0 is between unlink and create, or the run is over
      struct stat s2;
      const bool differs = fstat(f2, &s2) != 0 || s2.st_ino != ino;
      close(f2);
      return differs;
    };
    bool usable = false;
    for (int spin = 0; !expired(); ++spin) {
      if (seg->magic.load(std::memory_order_acquire) == kShmMagic) {
        usable = seg->failed.load() == 0 && seg->generation.load() == 0 && seg->attached.load() < c->world;
        break;  // initialised: usable, or a leftover of an earlier run
      }
      if (seg->failed.load() != 0 || ((spin & 63) == 63 && replaced())) break;  // poisoned or superseded
      usleep(200);
    }
    if (usable) {
      c->shm = seg;
      c->local_generation = 0;
      seg->attached.fetch_add(1);
      if (shm_barrier(c)) return GH_OK;
      c->shm = nullptr;
      // the barrier failed: a real peer failure of THIS run, or rank 0 has just poisoned a stale segment we were parked in
      const bool stale = seg->magic.load(std::memory_order_acquire) != kShmMagic || replaced();
      munmap(seg, sizeof(ShmSegment));
      if (!stale) return gh_set_error(ctx, GH_ERR_HIP, "IPC transport: rendezvous timed out (%d ranks expected)", c->world);
      continue;
    }
    munmap(seg, sizeof(ShmSegment));
    usleep(2000);  // stale segment: wait for rank 0 to replace it
  }
  return gh_set_error(ctx, GH_ERR_HIP, "IPC transport: no live rendezvous segment %s within %.0f s (is rank 0 running?)", name,
                      c->timeout_s);
}

void comm_common_free(gh_comm* c) {
  for (int p = 0; p < kMaxWorld; ++p) {
    if (c->pev[p]) hipEventDestroy(c->pev[p]);
    if (c->pstream[p]) hipStreamDestroy(c->pstream[p]);
    c->pev[p] = nullptr;
    c->pstream[p] = nullptr;
  }
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  c->ev_fork = nullptr;
  if (c->ev_ready) hipEventDestroy(c->ev_ready);
  if (c->ev_done) hipEventDestroy(c->ev_done);
  if (c->stream) hipStreamDestroy(c->stream);
  c->ev_ready = c->ev_done = nullptr;
  c->stream = nullptr;
}

IpcBuffer* find_ipc_buffer(gh_comm* c, const void* gathered, size_t bytes_per_rank) {
  for (auto& b : c->ipc_buffers)
    if (b.mine == gathered && bytes_per_rank <= b.bytes_per_rank) return &b;
  return nullptr;
}

// One gather of `bytes` per rank into `gathered` (world * stride bytes, rank r's slice at r * stride).
gh_status gather_one(gh_comm* c, const void* send, void* gathered, size_t bytes, size_t stride) {
  gh_ctx* ctx = c->ctx;
  if (c->transport == 0) {
    GH_CHECK_ARG(ctx, stride == bytes);  // ncclAllGather packs the slices back to back
    GH_NCCL(ctx, rccl().AllGather(send, gathered, bytes, ncclUint8, c->nccl, c->stream));
    return GH_OK;
  }
  IpcBuffer* b = find_ipc_buffer(c, gathered, stride);
  if (!b) return gh_set_error(ctx, GH_ERR_ARG, "IPC transport: the gathered buffer must come from gh_comm_buffer");
  const bool fan = c->peer_streams && c->ev_fork != nullptr;
  if (fan) GH_HIP(ctx, hipEventRecord(c->ev_fork, c->stream));  // behind the ready-poll of this exchange (and earlier gathers of it)
  for (int r = 0; r < c->world; ++r) {
    const int p = (c->rank + r) % c->world;  // stagger the targets so the ranks do not all hit the same peer first
    char* dst = (char*)b->peer[p] + (size_t)c->rank * stride;
    if (dst == (const char*)send) continue;  // in-place slice of my own buffer
    if (fan && p != c->rank) {
      GH_HIP(ctx, hipStreamWaitEvent(c->pstream[p], c->ev_fork, 0));
      GH_HIP(ctx, hipMemcpyAsync(dst, send, bytes, hipMemcpyDeviceToDevice, c->pstream[p]));
      GH_HIP(ctx, hipEventRecord(c->pev[p], c->pstream[p]));
      GH_HIP(ctx, hipStreamWaitEvent(c->stream, c->pev[p], 0));  // the done flag (end_collective) comes behind every push
    } else {
      GH_HIP(ctx, hipMemcpyAsync(dst, send, bytes, hipMemcpyDeviceToDevice, c->stream));
    }
  }
  return GH_OK;
}

gh_status begin_collective(gh_comm* c) {
  gh_ctx* ctx = c->ctx;
  if (c->pending) return gh_set_error(ctx, GH_ERR_ARG, "gh_comm_wait must be called before the next gather");
  if (c->transport == 1 && c->ipc_sync) {
    // peers still read their gathered buffers of the previous exchange through kernels on THEIR streams, and my send
    // data is produced on mine: drain, then meet -- after the barrier every buffer may be overwritten
    GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!shm_barrier(c)) return gh_set_error(ctx, GH_ERR_HIP, "IPC transport: a peer did not reach the barrier");
  } else if (c->transport == 1) {
    if (c->shm->failed.load(std::memory_order_relaxed))
      return gh_set_error(ctx, GH_ERR_HIP, "IPC transport: a rank gave up waiting in an earlier exchange (timeout %.0f s)", c->timeout_s);
    const uint32_t k = ++c->round;
    // ready[rank] = k once everything before this call has run on my context stream ...
    hipLaunchKernelGGL(comm_flag_set_kernel, dim3(1), dim3(1), 0, ctx->stream, &c->shm_dev->ready[c->rank].v, k);
    // ... and my communicator stream pushes when EVERY rank (me included: my send data) has got there
    hipLaunchKernelGGL(comm_flag_wait_kernel, dim3(1), dim3(64), 0, c->stream, c->shm_dev->ready, c->world, k,
                       (unsigned long long)(c->timeout_s * 1e8), (int*)&c->shm_dev->failed);
    GH_HIP(ctx, hipGetLastError());
  } else {
    GH_HIP(ctx, hipEventRecord(c->ev_ready, ctx->stream));
    GH_HIP(ctx, hipStreamWaitEvent(c->stream, c->ev_ready, 0));
  }
  return GH_OK;
}

// A rank that has announced an exchange (ready[rank] = k) and then fails locally (a buffer that is not a communicator
// buffer, a stride mismatch) will never publish done[rank] = k: without this its peers would sit out the whole timeout.
// The communicator is poisoned at once -- every poll stops, every later call on every rank reports it.
gh_status abandon_collective(gh_comm* c, gh_status st) {
  if (c->transport == 1 && !c->ipc_sync && c->shm) c->shm->failed.store(1, std::memory_order_relaxed);
  return st;
}

gh_status end_collective(gh_comm* c) {
  if (c->transport == 1 && !c->ipc_sync) {
    hipLaunchKernelGGL(comm_flag_set_kernel, dim3(1), dim3(1), 0, c->stream, &c->shm_dev->done[c->rank].v, c->round);
    GH_HIP(c->ctx, hipGetLastError());
  } else {
    GH_HIP(c->ctx, hipEventRecord(c->ev_done, c->stream));
  }
  c->pending = true;
  return GH_OK;
}

}  // namespace

extern "C" gh_status gh_comm_unique_id(uint8_t id_out[128]) {
  if (!id_out) return GH_ERR_ARG;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  if (!rccl().ok) return GH_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (rccl().GetUniqueId(&id) != ncclSuccess) return GH_ERR_HIP;
  memcpy(id_out, &id, 128);
  return GH_OK;
}

extern "C" gh_status gh_comm_create_rccl(gh_ctx* ctx, int rank, int world, const uint8_t unique_id[128], gh_comm** out) {
  if (!ctx || !out) return GH_ERR_ARG;
  GH_ENTER(ctx);
  *out = nullptr;
  GH_CHECK_ARG(ctx, world >= 1 && rank >= 0 && rank < world && unique_id);
  if (!rccl().ok) {
    const char* why = dlerror();
    return gh_set_error(ctx, GH_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded or lacks a required symbol: %s",
                        why ? why : "(no loader message)");
  }
  gh_comm* c = new (std::nothrow) gh_comm();
  if (!c) return GH_ERR_NOMEM;
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  c->transport = 0;
  gh_status st = comm_common_init(ctx, c);
  if (st == GH_OK) {
    ncclUniqueId id;
    memcpy(&id, unique_id, 128);
    ncclResult_t r = rccl().CommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) st = rccl_fail(ctx, "ncclCommInitRank", r);
  }
  if (st != GH_OK) {
    comm_common_free(c);
    delete c;
    return st;
  }
  *out = c;
  return GH_OK;
}

extern "C" gh_status gh_comm_create_ipc(gh_ctx* ctx, int rank, int world, const char* rendezvous_name, gh_comm** out) {
  if (!ctx || !out) return GH_ERR_ARG;
  GH_ENTER(ctx);
  *out = nullptr;
  GH_CHECK_ARG(ctx, world >= 1 && world <= kMaxWorld && rank >= 0 && rank < world && rendezvous_name && rendezvous_name[0]);
  gh_comm* c = new (std::nothrow) gh_comm();
  if (!c) return GH_ERR_NOMEM;
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  c->transport = 1;
  c->shm_name = std::string(rendezvous_name[0] == '/' ? "" : "/") + rendezvous_name;
  if (const char* t = getenv("GSLAM_HIP_COMM_TIMEOUT_S")) c->timeout_s = atof(t) > 0 ? atof(t) : c->timeout_s;
  if (const char* e = getenv("GSLAM_HIP_IPC_SYNC")) c->ipc_sync = atoi(e) != 0;
  c->peer_streams = world > 2;  // (with one peer there is one link: the fork / join events would only cost host time)
  if (const char* e = getenv("GSLAM_HIP_IPC_PEER_STREAMS")) c->peer_streams = atoi(e) != 0;  // 1 forces them on (2-rank tests)
  gh_status st = comm_common_init(ctx, c);
  if (st == GH_OK && c->peer_streams) {
    bool ok = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int p = 0; ok && p < world; ++p) {
      if (p == rank) continue;
      ok = hipStreamCreateWithFlags(&c->pstream[p], hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&c->pev[p], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) st = gh_set_error(ctx, GH_ERR_HIP, "IPC transport: cannot create the per-peer streams");
  }
  if (st == GH_OK) st = ipc_rendezvous(ctx, c);
  if (st == GH_OK && !c->ipc_sync) {
    // the GPU's view of the segment (pinned + mapped: device loads / stores go straight to host memory)
    hipError_t e = hipHostRegister(c->shm, sizeof(ShmSegment), hipHostRegisterMapped);
    if (e == hipSuccess) {
      c->shm_registered = true;
      e = hipHostGetDevicePointer((void**)&c->shm_dev, c->shm, 0);
    }
    if (e != hipSuccess) {
      c->shm->failed.store(1);
      st = gh_set_error(ctx, GH_ERR_HIP, "IPC transport: cannot map the rendezvous segment into the GPU: %s", hipGetErrorString(e));
    }
  }
  if (st != GH_OK) {
    if (c->shm_registered) hipHostUnregister(c->shm);
    if (c->shm) munmap(c->shm, sizeof(ShmSegment));
    comm_common_free(c);
    delete c;
    return st;
  }
  *out = c;
  return GH_OK;
}

extern "C" void gh_comm_destroy(gh_comm* c) {
  if (!c) return;
  GH_ENTER(c->ctx);
  hipStreamSynchronize(c->stream);
  hipStreamSynchronize(c->ctx->stream);
  if (c->transport == 0) {
    if (c->nccl) rccl().CommDestroy(c->nccl);
    for (void* p : c->plain_buffers) hipFree(p);
  } else {
    // nobody may unmap a buffer a peer is still writing into
    if (c->shm && !c->shm->failed.load()) shm_barrier(c);
    for (auto& b : c->ipc_buffers) {
      for (int r = 0; r < c->world; ++r)
        if (r != c->rank && b.peer[r]) hipIpcCloseMemHandle(b.peer[r]);
    }
    if (c->shm && !c->shm->failed.load()) shm_barrier(c);
    for (auto& b : c->ipc_buffers) hipFree(b.mine);
    if (c->shm_registered) hipHostUnregister(c->shm);
    if (c->shm) {
      const int left = c->shm->attached.fetch_sub(1) - 1;
      munmap(c->shm, sizeof(ShmSegment));
      if (left == 0 || c->rank == 0) shm_unlink(c->shm_name.c_str());
    }
  }
  for (int p = 0; p < kMaxWorld; ++p) {
    if (c->pev[p]) hipEventDestroy(c->pev[p]);
    if (c->pstream[p]) hipStreamDestroy(c->pstream[p]);
  }
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  hipEventDestroy(c->ev_ready);
  hipEventDestroy(c->ev_done);
  hipStreamDestroy(c->stream);
  delete c;
}

// GH_OK, or the error of an exchange that a rank abandoned (a bounded poll ran out): the asynchronous IPC transport
// cannot report that from gh_comm_wait, which does not wait.  Cheap (one host load); call it after a stream sync.
extern "C" gh_status gh_comm_status(gh_comm* c) {
  if (!c) return GH_ERR_ARG;
  if (c->transport == 1 && c->shm && c->shm->failed.load(std::memory_order_relaxed))
    return gh_set_error(c->ctx, GH_ERR_HIP, "IPC transport: a rank gave up waiting for its peers (timeout %.0f s)", c->timeout_s);
  return GH_OK;
}

extern "C" int gh_comm_rank(const gh_comm* c) { return c ? c->rank : -1; }
extern "C" int gh_comm_world(const gh_comm* c) { return c ? c->world : 0; }

// Collective: every rank calls it with the same size.  Returns a device buffer of world * bytes_per_rank bytes owned by
// the communicator (freed by gh_comm_destroy); with the IPC transport every peer's copy is mapped into this process.
extern "C" gh_status gh_comm_buffer(gh_comm* c, size_t bytes_per_rank, void** gathered_dev) {
  if (!c || !gathered_dev) return GH_ERR_ARG;
  gh_ctx* ctx = c->ctx;
  GH_ENTER(ctx);
  *gathered_dev = nullptr;
  GH_CHECK_ARG(ctx, bytes_per_rank > 0);
  const size_t total = bytes_per_rank * (size_t)c->world;
  void* p = nullptr;
  if (hipMalloc(&p, total) != hipSuccess) return gh_set_error(ctx, GH_ERR_NOMEM, "hipMalloc(%zu) for a gathered buffer", total);
  if (c->transport == 0) {
    c->plain_buffers.push_back(p);
    *gathered_dev = p;
    return GH_OK;
  }
  if ((int)c->ipc_buffers.size() >= kMaxBuffers) {
    hipFree(p);
    return gh_set_error(ctx, GH_ERR_ARG, "IPC transport: at most %d gathered buffers per communicator", kMaxBuffers);
  }
  IpcBuffer b;
  b.mine = p;
  b.bytes_per_rank = bytes_per_rank;
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) {
    c->shm->failed.store(1);
    hipFree(p);
    return gh_set_error(ctx, GH_ERR_HIP, "hipIpcGetMemHandle: %s (is HSA_ENABLE_IPC_MODE_LEGACY=0 exported?)", hipGetErrorString(e));
  }
  c->shm->handles[c->rank] = h;
  c->shm->sizes[c->rank] = total;
  bool ok = shm_barrier(c);  // every handle is published
  for (int r = 0; ok && r < c->world; ++r) {
    if (r == c->rank) {
      b.peer[r] = p;
      continue;
    }
    if (c->shm->sizes[r] != total) {
      ok = false;
      gh_set_error(ctx, GH_ERR_ARG, "gh_comm_buffer: rank %d asked for %llu bytes, this rank for %zu", r,
                   (unsigned long long)c->shm->sizes[r], total);
      break;
    }
    e = hipIpcOpenMemHandle(&b.peer[r], c->shm->handles[r], hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      ok = false;
      gh_set_error(ctx, GH_ERR_HIP, "hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e));
    }
  }
  if (!ok) c->shm->failed.store(1);
  if (!shm_barrier(c) || !ok) {  // the handle slots may be reused only after everybody has opened them
    if (ok) gh_set_error(ctx, GH_ERR_HIP, "IPC transport: a peer failed while exchanging buffer handles");
    for (int r = 0; r < c->world; ++r)
      if (r != c->rank && b.peer[r]) hipIpcCloseMemHandle(b.peer[r]);
    hipFree(p);
    return GH_ERR_HIP;
  }
  c->ipc_buffers.push_back(b);
  *gathered_dev = p;
  return GH_OK;
}

extern "C" gh_status gh_allgather(gh_comm* c, const void* send_dev, void* gathered_dev, size_t bytes_per_rank) {
  if (!c) return GH_ERR_ARG;
  gh_ctx* ctx = c->ctx;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, send_dev && gathered_dev && bytes_per_rank > 0);
  GH_TRY(begin_collective(c));
  const gh_status st = gather_one(c, send_dev, gathered_dev, bytes_per_rank, bytes_per_rank);
  if (st != GH_OK) return abandon_collective(c, st);
  return end_collective(c);
}

// The exchange after extraction (SURVEY.md 8e record {n, KeyPoint[K], desc[K][32]} per frame, kept as the three arrays
// gh_orb_extract_dev writes): this rank's `frames` frames into slot `rank` of the three gathered arrays.  kps may be
// NULL on every rank when only descriptors are needed (consecutive-pair matching); the stereo band matcher needs them.
extern "C" gh_status gh_allgather_features(gh_comm* c, int frames, int cap, const gh_keypoint* kps_dev,
                                           const uint8_t* desc_dev, const int32_t* counts_dev, gh_keypoint* g_kps_dev,
                                           uint8_t* g_desc_dev, int32_t* g_counts_dev) {
  if (!c) return GH_ERR_ARG;
  gh_ctx* ctx = c->ctx;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, frames > 0 && cap > 0 && desc_dev && counts_dev && g_desc_dev && g_counts_dev);
  GH_CHECK_ARG(ctx, (kps_dev == nullptr) == (g_kps_dev == nullptr));
  GH_TRY(begin_collective(c));
  if (c->transport == 0) GH_NCCL(ctx, rccl().GroupStart());
  gh_status st = gather_one(c, desc_dev, g_desc_dev, (size_t)frames * cap * 32, (size_t)frames * cap * 32);
  if (st == GH_OK) st = gather_one(c, counts_dev, g_counts_dev, (size_t)frames * 4, (size_t)frames * 4);
  if (st == GH_OK && kps_dev)
    st = gather_one(c, kps_dev, g_kps_dev, (size_t)frames * cap * sizeof(gh_keypoint), (size_t)frames * cap * sizeof(gh_keypoint));
  if (c->transport == 0) {
    ncclResult_t r = rccl().GroupEnd();
    if (st == GH_OK && r != ncclSuccess) st = rccl_fail(ctx, "ncclGroupEnd", r);
  }
  if (st != GH_OK) return abandon_collective(c, st);
  return end_collective(c);
}

// The exchange after matching: `rows` match rows (cap int32 train indices + optional cap u16 best / second distances).
extern "C" gh_status gh_allgather_matches(gh_comm* c, int rows, int cap, const int32_t* idx1_dev, const uint16_t* d1_dev,
                                          const uint16_t* d2_dev, int32_t* g_idx1_dev, uint16_t* g_d1_dev,
                                          uint16_t* g_d2_dev) {
  if (!c) return GH_ERR_ARG;
  gh_ctx* ctx = c->ctx;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, rows > 0 && cap > 0 && idx1_dev && g_idx1_dev);
  GH_CHECK_ARG(ctx, (d1_dev == nullptr) == (g_d1_dev == nullptr) && (d2_dev == nullptr) == (g_d2_dev == nullptr));
  GH_TRY(begin_collective(c));
  if (c->transport == 0) GH_NCCL(ctx, rccl().GroupStart());
  gh_status st = gather_one(c, idx1_dev, g_idx1_dev, (size_t)rows * cap * 4, (size_t)rows * cap * 4);
  if (st == GH_OK && d1_dev) st = gather_one(c, d1_dev, g_d1_dev, (size_t)rows * cap * 2, (size_t)rows * cap * 2);
  if (st == GH_OK && d2_dev) st = gather_one(c, d2_dev, g_d2_dev, (size_t)rows * cap * 2, (size_t)rows * cap * 2);
  if (c->transport == 0) {
    ncclResult_t r = rccl().GroupEnd();
    if (st == GH_OK && r != ncclSuccess) st = rccl_fail(ctx, "ncclGroupEnd", r);
  }
  if (st != GH_OK) return abandon_collective(c, st);
  return end_collective(c);
}

// Orders the context's stream after the gather in flight; the host does not wait (with GSLAM_HIP_IPC_SYNC=1 the IPC
// transport blocks until every rank's slice has landed).  A no-op when nothing is pending.
extern "C" gh_status gh_comm_wait(gh_comm* c) {
  if (!c) return GH_ERR_ARG;
  gh_ctx* ctx = c->ctx;
  GH_ENTER(ctx);
  if (!c->pending) return GH_OK;
  c->pending = false;
  if (c->transport == 1 && c->ipc_sync) {
    GH_HIP(ctx, hipStreamSynchronize(c->stream));  // my pushes have landed everywhere ...
    if (!shm_barrier(c)) return gh_set_error(ctx, GH_ERR_HIP, "IPC transport: a peer did not finish its pushes");
    return GH_OK;  // ... and so have everybody else's
  }
  if (c->transport == 1) {
    // the context stream goes on when every rank's pushes of this exchange have landed (mine included)
    hipLaunchKernelGGL(comm_flag_wait_kernel, dim3(1), dim3(64), 0, ctx->stream, c->shm_dev->done, c->world, c->round,
                       (unsigned long long)(c->timeout_s * 1e8), (int*)&c->shm_dev->failed);
    GH_HIP(ctx, hipGetLastError());
    // This call does not wait, so a peer that goes missing in THIS exchange shows only once the poll above has run out:
    // in gh_comm_status (call it after the next stream synchronisation -- bench.py and gslam_amd/sharding.py do) and in
    // every later gh_comm_* call, this one included: an exchange that was abandoned earlier is reported here
    if (c->shm->failed.load(std::memory_order_relaxed))
      return gh_set_error(ctx, GH_ERR_HIP, "IPC transport: a rank gave up waiting in an earlier exchange (timeout %.0f s): "
                                           "the gathered buffers are incomplete", c->timeout_s);
    return GH_OK;
  }
  GH_HIP(ctx, hipStreamWaitEvent(ctx->stream, c->ev_done, 0));
  return GH_OK;
}
