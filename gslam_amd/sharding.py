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
