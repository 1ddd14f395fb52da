"""Multi-GPU sharding of the front end: frames are independent units, so rank r of a
world of G owns the contiguous block of global frames [r F, (r+1) F).  There is exactly one exchange
step on the data path: an RCCL all-gather of the fixed-capacity per-frame records (descriptors K x 32 B
+ count) so that every GPU holds every frame's descriptors, optionally followed by an all-gather of
the match records.  Pure data movement: gathered buffers are byte-identical for every G.
Bundle adjustment does not shard (one coupled dense system): replicas only.

torch.distributed (backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests) is plumbing here.
"""
import torch
import torch.distributed as dist


def local_pairs(rank: int, world: int, frames_per_rank: int, device="cpu"):
    """Consecutive global frame pairs (g, g+1) owned by `rank`: g in [rank F, (rank+1) F), g+1 < world F.
    Returned as indices into the gathered (world*F) frame axis."""
    g0 = rank * frames_per_rank
    g1 = min((rank + 1) * frames_per_rank, world * frames_per_rank - 1)
    pq = torch.arange(g0, g1, dtype=torch.int32, device=device)
    return pq, pq + 1


def all_pairs_block(rank: int, world: int, n_frames_total: int, device="cpu"):
    """Upper-triangle all-pairs (i < j) of the gathered frames, dealt round-robin to ranks."""
    i, j = torch.triu_indices(n_frames_total, n_frames_total, offset=1)
    sel = torch.arange(i.numel()) % world == rank
    return i[sel].to(torch.int32).to(device), j[sel].to(torch.int32).to(device)


def _all_gather_flat(out: torch.Tensor, inp: torch.Tensor):
    """all_gather_into_tensor; device tensors over a CPU-only backend (gloo, used by the 1-GPU dry run of the
    multi-rank code path) are staged through host memory."""
    if out.is_cuda and dist.get_backend() == "gloo":
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu())
        out.copy_(o)
    else:
        dist.all_gather_into_tensor(out, inp)


class _Pending:
    """Handle of an all-gather in flight (see exchange_features_begin)."""

    def __init__(self, works, finish):
        self.works, self.finish = works, finish

    def wait(self):
        for w in self.works:
            w.wait()  # nccl: the CURRENT STREAM waits (the host does not block); gloo: the host waits
        for f in self.finish:
            f()
        self.works, self.finish = [], []


def _all_gather_begin(out: torch.Tensor, inp: torch.Tensor):
    if out.is_cuda and dist.get_backend() == "gloo":  # CPU-only backend: stage through host memory
        o = torch.empty(out.shape, dtype=out.dtype)
        w = dist.all_gather_into_tensor(o, inp.cpu(), async_op=True)
        return w, (lambda: out.copy_(o))
    return dist.all_gather_into_tensor(out, inp, async_op=True), None


def exchange_features_begin(desc: torch.Tensor, counts: torch.Tensor, g_desc: torch.Tensor, g_counts: torch.Tensor):
    """Start the all-gather of (F,K,32) u8 descriptors and (F,) counts into (G*F,K,32) / (G*F,).  With RCCL the
    collective runs on its own stream (ordered after the work already queued on the current one), so kernels launched
    before .wait() -- the matching of the rank's own frame pairs -- overlap the transfer over xGMI."""
    works, finish = [], []
    for o, i in ((g_desc.view(-1), desc.contiguous().view(-1)), (g_counts, counts)):
        w, f = _all_gather_begin(o, i)
        works.append(w)
        if f is not None:
            finish.append(f)
    return _Pending(works, finish)


def exchange_features(desc: torch.Tensor, counts: torch.Tensor, g_desc: torch.Tensor, g_counts: torch.Tensor):
    """all-gather (F,K,32) u8 descriptors and (F,) counts into (G*F,K,32) / (G*F,) (blocking form)."""
    exchange_features_begin(desc, counts, g_desc, g_counts).wait()


def exchange_matches_begin(idx1: torch.Tensor, g_idx1: torch.Tensor, frames_per_rank: int):
    """Start the all-gather of per-rank (P,K) int32 match rows into (G,F,K); ranks with fewer than F pairs pad with -1.
    The send buffer is a private copy, so the caller may overwrite idx1 (next step) while the transfer is in flight."""
    P, K = idx1.shape
    if P < frames_per_rank:
        pad = torch.full((frames_per_rank - P, K), -1, dtype=idx1.dtype, device=idx1.device)
        send = torch.cat([idx1, pad])
    else:
        send = idx1.clone()
    w, f = _all_gather_begin(g_idx1.view(-1), send.contiguous().view(-1))
    p = _Pending([w], [f] if f is not None else [])
    p.keepalive = send
    return p


def exchange_matches(idx1: torch.Tensor, g_idx1: torch.Tensor, frames_per_rank: int):
    """Blocking form of exchange_matches_begin."""
    exchange_matches_begin(idx1, g_idx1, frames_per_rank).wait()


# ---------------------------------------------------------------------------------------------------------------------
# The same exchange through the C ABI (include/gslam_hip.h, gh_comm_*): what a C++ host uses, and what bench.py / the
# tests use from here on.  torch only supplies the views of the gathered buffers and (for RCCL) carries the 128-byte
# unique id from rank 0 to the other ranks.
import ctypes as _C

from . import hip as _hip


class _DeviceView:
    """Zero-copy torch view of device memory owned by the communicator (CUDA array interface)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class Comm:
    """gh_comm: RCCL all-gather over xGMI, or same-node HIP-IPC direct writes (several ranks may share a GPU)."""

    def __init__(self, ctx, handle, rank, world, transport):
        self.ctx, self.h, self.rank, self.world, self.transport = ctx, handle, rank, world, transport
        self._keep = []

    @classmethod
    def rccl(cls, ctx, rank, world):
        """Bootstrap: rank 0 creates the id, torch.distributed (any backend) broadcasts its 128 bytes."""
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (_C.c_uint8 * 128)()
            st = _hip.lib.gh_comm_unique_id(buf)
            if st != 0:
                raise _hip.GslamHipError(f"gh_comm_unique_id failed with status {st} (librccl.so.1 not loadable?)")
            uid = torch.frombuffer(bytearray(buf), dtype=torch.uint8).clone()
        if world > 1:
            dev = uid.cuda() if dist.get_backend() == "nccl" else uid
            dist.broadcast(dev, src=0)
            uid = dev.cpu()
        raw = (_C.c_uint8 * 128)(*uid.tolist())
        h = _C.c_void_p()
        ctx.check(_hip.lib.gh_comm_create_rccl(ctx.h, rank, world, raw, _C.byref(h)))
        return cls(ctx, h, rank, world, "rccl")

    @classmethod
    def ipc(cls, ctx, rank, world, name):
        h = _C.c_void_p()
        ctx.check(_hip.lib.gh_comm_create_ipc(ctx.h, rank, world, name.encode(), _C.byref(h)))
        return cls(ctx, h, rank, world, "ipc")

    def close(self):
        if self.h:
            self._keep = []
            _hip.lib.gh_comm_destroy(self.h)
            self.h = None

    def buffer(self, per_rank_shape, dtype):
        """Collective.  Gathered buffer of shape (world, *per_rank_shape) as a torch view."""
        n = int(torch.tensor(per_rank_shape).prod().item()) * torch.empty(0, dtype=dtype).element_size()
        p = _C.c_void_p()
        self.ctx.check(_hip.lib.gh_comm_buffer(self.h, n, _C.byref(p)))
        view = _DeviceView(p.value, n * self.world)
        t = torch.as_tensor(view, device="cuda").view(dtype).view(self.world, *per_rank_shape)
        assert t.data_ptr() == p.value
        self._keep.append(view)
        return t

    def allgather_features(self, desc, counts, g_desc, g_counts, kps=None, g_kps=None):
        F, K = desc.shape[0], desc.shape[1]
        p = lambda t: _C.c_void_p(t.data_ptr()) if t is not None else None
        self.ctx.check(_hip.lib.gh_allgather_features(self.h, F, K, p(kps), p(desc), p(counts), p(g_kps), p(g_desc), p(g_counts)))

    def allgather_matches(self, idx1, g_idx1, d1=None, g_d1=None, d2=None, g_d2=None):
        R, K = idx1.shape
        p = lambda t: _C.c_void_p(t.data_ptr()) if t is not None else None
        self.ctx.check(_hip.lib.gh_allgather_matches(self.h, R, K, p(idx1), p(d1), p(d2), p(g_idx1), p(g_d1), p(g_d2)))

    def allgather(self, send, gathered):
        self.ctx.check(_hip.lib.gh_allgather(self.h, _C.c_void_p(send.data_ptr()), _C.c_void_p(gathered.data_ptr()),
                                             send.numel() * send.element_size()))

    def wait(self):
        """Orders the context's stream after the gather in flight; the host does not wait (both transports)."""
        self.ctx.check(_hip.lib.gh_comm_wait(self.h))

    def status(self):
        """Raises once any rank has abandoned an exchange (bounded polls of the IPC transport); call after a sync."""
        self.ctx.check(_hip.lib.gh_comm_status(self.h))
