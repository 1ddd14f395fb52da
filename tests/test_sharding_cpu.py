"""N > 1 path on CPU: world_size-2 gloo processes exercise the frame sharding + all-gather exchange
(the same functions bench.py runs over RCCL) with the oracle standing in for the GPU kernels."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, F, K, out_dir):
    import oracle_lib
    from gslam_amd.sharding import exchange_features_begin, exchange_matches_begin, local_pairs
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # "extract": per-frame descriptor sets with ragged counts, a pure function of the global frame index
    desc = torch.zeros((F, K, 32), dtype=torch.uint8)
    counts = torch.zeros(F, dtype=torch.int32)
    for f in range(F):
        g = rank * F + f
        n = K - (g * 7) % 13
        desc[f, :n] = torch.from_numpy(oracle_lib.random_descriptors(n, 1000 + g))
        counts[f] = n
    g_desc = torch.empty((world * F, K, 32), dtype=torch.uint8)
    g_counts = torch.empty(world * F, dtype=torch.int32)
    # the schedule bench.py uses: start the gather, match the purely local pairs from the LOCAL buffers while it is in
    # flight, wait, then the boundary pair from the gathered buffers; the match rows travel asynchronously too
    pending = exchange_features_begin(desc, counts, g_desc, g_counts)
    pq, pt = local_pairs(rank, world, F)
    oracle = oracle_lib.load()
    idx1 = torch.full((pq.numel(), K), -1, dtype=torch.int32)
    n_local = min(F - 1, pq.numel())
    for p in range(n_local):
        na, nb = int(counts[p]), int(counts[p + 1])
        e = oracle.bf_match(desc[p, :na].numpy(), desc[p + 1, :nb].numpy())
        idx1[p, :na] = torch.from_numpy(e[0])
    pending.wait()
    for p in range(n_local, pq.numel()):
        a, b = int(pq[p]), int(pt[p])
        na, nb = int(g_counts[a]), int(g_counts[b])
        e = oracle.bf_match(g_desc[a, :na].numpy(), g_desc[b, :nb].numpy())
        idx1[p, :na] = torch.from_numpy(e[0])
    g_idx = torch.empty((world, F, K), dtype=torch.int32)
    mg = exchange_matches_begin(idx1, g_idx, F)
    idx1.fill_(-7)  # the send buffer is a private copy: the caller may reuse idx1 at once
    mg.wait()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), g_desc=g_desc.numpy(), g_counts=g_counts.numpy(),
             g_idx=g_idx.numpy(), pq=pq.numpy())
    dist.destroy_process_group()


def test_two_rank_gloo_exchange_matches_single_rank(tmp_path):
    import oracle_lib
    world, F, K = 2, 3, 40
    port = _free_port()
    mp.spawn(_worker, args=(world, port, F, K, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # gathered buffers are byte-identical on every rank
    for k in ("g_desc", "g_counts", "g_idx"):
        assert np.array_equal(r0[k], r1[k])
    # pair ownership: every consecutive global pair exactly once
    allq = np.concatenate([r0["pq"], r1["pq"]])
    assert sorted(allq.tolist()) == list(range(world * F - 1))
    # and equal to the single-rank computation over the same 6 frames
    oracle = oracle_lib.load()
    g_idx = r0["g_idx"].reshape(world * F, K)
    for g in range(world * F - 1):
        na, nb = K - (g * 7) % 13, K - ((g + 1) * 7) % 13
        a = oracle_lib.random_descriptors(na, 1000 + g)
        b = oracle_lib.random_descriptors(nb, 1000 + g + 1)
        assert np.array_equal(r0["g_desc"][g, :na], a)
        e = oracle.bf_match(a, b)
        assert np.array_equal(g_idx[g, :na], e[0])
    assert (g_idx[world * F - 1] == -1).all()  # the last global frame has no successor


def test_local_pairs_and_all_pairs_partition():
    from gslam_amd.sharding import all_pairs_block, local_pairs
    for world in (1, 2, 4, 8):
        F = 5
        seen = []
        for r in range(world):
            pq, pt = local_pairs(r, world, F)
            assert torch.equal(pt, pq + 1)
            seen += pq.tolist()
        assert sorted(seen) == list(range(world * F - 1))
        tot = 0
        pairs = set()
        for r in range(world):
            i, j = all_pairs_block(r, world, 7)
            tot += i.numel()
            pairs |= set(zip(i.tolist(), j.tolist()))
        assert tot == 21 and len(pairs) == 21
