"""Multi-GPU exchange behind the C ABI (gh_comm_*, gslam_amd/csrc/comm.hip) with the REAL kernels.

A 1-GPU box cannot host two RCCL ranks ("Duplicate GPU detected"), so the two-rank run uses the same-node IPC transport
(both processes on GPU 0, peers' buffers mapped through HIP IPC): extract -> gh_allgather_features (descriptors, counts
AND keypoints) -> consecutive-pair matching incl. the boundary pair -> stereo band matching on gathered keypoints ->
gh_allgather_matches.  Every gathered buffer must hash identically on both ranks and equal the world = 1 result of the
same frames.  The RCCL transport is exercised at world = 1 (the API path incl. dlopen of librccl.so.1)."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, {root!r})
from gslam_amd import hip
from gslam_amd.matcher import BFMatcher
from gslam_amd.orb import OrbExtractor, synth_frames
from gslam_amd.sharding import Comm, local_pairs

rank, world, F, K, W, H, name, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7], sys.argv[8]
torch.cuda.set_device(0)
ctx = hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)
comm = Comm.ipc(ctx, rank, world, name) if name != "rccl" else Comm.rccl(ctx, rank, world)
ex = OrbExtractor(ctx, W, H, max_batch=F, n_features=K)
frames = synth_frames(ctx, F, W, H, base_seed=0xC3000000, first_frame=rank * F)
kps, desc, counts = ex.extract(frames)
g_desc = comm.buffer((F, K, 32), torch.uint8)
g_counts = comm.buffer((F,), torch.int32)
g_kps = comm.buffer((F, K, 7), torch.float32)
m = BFMatcher(ctx)
res = {{}}
for step in range(2):  # twice: the second exchange overwrites buffers the peers have been reading
    comm.allgather_features(desc, counts, g_desc, g_counts, kps, g_kps)
    # overlapped with the gather: the purely local consecutive pairs
    n_local = F - 1
    lq = torch.arange(0, n_local, dtype=torch.int32, device="cuda")
    loc = m.match_pairs(desc, counts, lq, lq + 1)
    comm.wait()
    gd, gc, gk = g_desc.view(world * F, K, 32), g_counts.view(world * F), g_kps.view(world * F, K, 7)
    pq, pt = local_pairs(rank, world, F, "cuda")
    idx1 = torch.full((F, K), -1, dtype=torch.int32, device="cuda")
    d1 = torch.full((F, K), -1, dtype=torch.int16, device="cuda")
    d2 = torch.full((F, K), -1, dtype=torch.int16, device="cuda")
    P = pq.shape[0]
    idx1[:n_local], d1[:n_local], d2[:n_local] = loc
    if P > n_local:
        b = m.match_pairs(gd, gc, pq[n_local:], pt[n_local:])
        idx1[n_local:P], d1[n_local:P], d2[n_local:P] = b
    # band-limited matching (stereo-style) of frame g against g + 1 needs the gathered KEYPOINTS
    band = m.match_band_pairs(gd, gk, gc, pq, pt, 2.0 / 31.0)
    g_idx = comm.buffer((F, K), torch.int32) if step == 0 else g_idx
    g_d1 = comm.buffer((F, K), torch.int16) if step == 0 else g_d1
    g_d2 = comm.buffer((F, K), torch.int16) if step == 0 else g_d2
    comm.allgather_matches(idx1, g_idx, d1, g_d1, d2, g_d2)
    comm.wait()
    torch.cuda.synchronize()
    res[step] = dict(g_desc=gd.cpu().numpy(), g_counts=gc.cpu().numpy(), g_kps=gk.cpu().numpy(), g_idx=g_idx.cpu().numpy(),
                     g_d1=g_d1.cpu().numpy(), g_d2=g_d2.cpu().numpy(), band=band[0].cpu().numpy())
for k in res[0]:
    assert res[0][k].tobytes() == res[1][k].tobytes(), k
np.savez(out, **res[1])
comm.close(); ex.close(); ctx.close()
print("rank", rank, "ok")
'''


def _spawn(world, F, K, W, H, name, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GSLAM_HIP_COMM_TIMEOUT_S="60")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(F), str(K), str(W), str(H), name,
                               str(tmp_path / f"rank{r}_{name}.npz")], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r}:\n{outs[r][-3000:]}"
    return [np.load(tmp_path / f"rank{r}_{name}.npz") for r in range(world)]


def test_two_ranks_on_one_gpu_ipc_transport_real_kernels(tmp_path):
    F, K, W, H = 3, 600, 640, 376
    name = "gslam_comm_test_%d" % os.getpid()
    r0, r1 = _spawn(2, F, K, W, H, name, tmp_path)
    for k in ("g_desc", "g_counts", "g_kps", "g_idx", "g_d1", "g_d2"):
        assert hashlib.sha256(r0[k].tobytes()).hexdigest() == hashlib.sha256(r1[k].tobytes()).hexdigest(), k
    # world = 1 over the same 6 frames, in this process
    import torch
    from gslam_amd import hip
    from gslam_amd.matcher import BFMatcher
    from gslam_amd.orb import OrbExtractor, synth_frames
    torch.cuda.set_device(0)
    ctx = hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    ex = OrbExtractor(ctx, W, H, max_batch=2 * F, n_features=K)
    kps, desc, counts = ex.extract(synth_frames(ctx, 2 * F, W, H, base_seed=0xC3000000))
    q = torch.arange(0, 2 * F - 1, dtype=torch.int32, device="cuda")
    idx1, d1, d2 = BFMatcher(ctx).match_pairs(desc, counts, q, q + 1)
    band = BFMatcher(ctx).match_band_pairs(desc, kps, counts, q, q + 1, 2.0 / 31.0)
    torch.cuda.synchronize()
    assert np.array_equal(r0["g_desc"], desc.cpu().numpy()) and np.array_equal(r0["g_counts"], counts.cpu().numpy())
    assert r0["g_kps"].tobytes() == kps.cpu().numpy().tobytes()
    g_idx = r0["g_idx"].reshape(2 * F, K)
    assert np.array_equal(g_idx[:2 * F - 1], idx1.cpu().numpy()) and (g_idx[2 * F - 1] == -1).all()
    assert np.array_equal(r0["g_d1"].reshape(2 * F, K)[:2 * F - 1], d1.cpu().numpy())
    assert np.array_equal(r0["g_d2"].reshape(2 * F, K)[:2 * F - 1], d2.cpu().numpy())
    assert np.array_equal(np.concatenate([r0["band"], r1["band"]]), band[0].cpu().numpy())
    assert (idx1.cpu().numpy() >= 0).mean() > 0.5
    ex.close()
    ctx.close()


def test_rccl_transport_world_one(tmp_path):
    """The RCCL code path end to end at world = 1 (dlopen, ncclCommInitRank, grouped ncclAllGather on the communicator's
    stream, stream ordering through gh_comm_wait).  N > 1 needs N GPUs: the driver's scaling run."""
    (r0,) = _spawn(1, 3, 500, 640, 376, "rccl", tmp_path)
    assert (r0["g_counts"] > 0).all() and r0["g_desc"].any() and (r0["g_idx"][:2] >= 0).any()


def test_ipc_rendezvous_survives_a_stale_segment_and_late_rank0(tmp_path):
    """a crashed run leaves /dev/shm/<name> behind with its magic set and its barrier generation advanced; a rank
    >= 1 that starts before rank 0 finds it.  It must neither attach to it (its barriers would fall through) nor give up:
    rank 0 poisons + replaces the segment and the late rendezvous completes; results equal the clean run."""
    import struct
    import time
    F, K, W, H = 2, 300, 640, 376
    name = "gslam_comm_stale_%d" % os.getpid()
    # ShmSegment starts with magic (u32), arrived, generation, attached, failed (i32 each); the rest is handle space
    stale = struct.pack("<Iiiii", 0x47534C4D, 0, 7, 2, 0) + b"\0" * (1 << 16)
    with open("/dev/shm/" + name, "wb") as f:
        f.write(stale)
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GSLAM_HIP_COMM_TIMEOUT_S="60")

    def start(r):
        return subprocess.Popen([sys.executable, str(script), str(r), "2", str(F), str(K), str(W), str(H), name,
                                 str(tmp_path / f"rank{r}_{name}.npz")], env=env, stdout=subprocess.PIPE,
                                stderr=subprocess.STDOUT, text=True)
    p1 = start(1)
    time.sleep(6.0)  # rank 1 is well inside its rendezvous loop, looking at the stale segment
    assert p1.poll() is None, p1.communicate()[0][-2000:]
    p0 = start(0)
    outs = []
    for p in (p0, p1):
        try:
            outs.append(p.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            p0.kill(); p1.kill()
            raise
    assert p0.returncode == 0 and p1.returncode == 0, outs[0][-2000:] + outs[1][-2000:]
    r0, r1 = (np.load(tmp_path / f"rank{r}_{name}.npz") for r in range(2))
    for k in ("g_desc", "g_counts", "g_kps", "g_idx"):
        assert r0[k].tobytes() == r1[k].tobytes(), k
    assert (r0["g_counts"] > 0).all()
    assert not os.path.exists("/dev/shm/" + name)  # the last rank out unlinks it


ASYNC_WORKER = r'''
import os, sys, time
import numpy as np, torch
sys.path.insert(0, {root!r})
from gslam_amd import hip
from gslam_amd.matcher import BFMatcher
from gslam_amd.sharding import Comm, all_pairs_block

rank, world, mode, name = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
torch.cuda.set_device(0)
ctx = hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)
comm = Comm.ipc(ctx, rank, world, name)
F, K = 48, 1000
g = torch.Generator(device="cuda").manual_seed(7 + rank)
desc = torch.randint(0, 256, (F, K, 32), dtype=torch.uint8, device="cuda", generator=g)
counts = torch.full((F,), K, dtype=torch.int32, device="cuda")
g_desc = comm.buffer((F, K, 32), torch.uint8)
g_counts = comm.buffer((F,), torch.int32)
m = BFMatcher(ctx)
ai, aj = all_pairs_block(0, 1, F, "cuda")
out = m.match_pairs(desc, counts, ai, aj)
torch.cuda.synchronize()
if mode == "overlap":
    # how long the queued work takes on the GPU
    t0 = time.perf_counter()
    for _ in range(4):
        m.match_pairs(desc, counts, ai, aj, out=out)
    torch.cuda.synchronize()
    busy = time.perf_counter() - t0
    # the same work queued again, then an exchange behind it: the host must come back long before the GPU is done
    for _ in range(4):
        m.match_pairs(desc, counts, ai, aj, out=out)
    t0 = time.perf_counter()
    comm.allgather_features(desc, counts, g_desc, g_counts)
    comm.wait()
    host = time.perf_counter() - t0
    after = m.match_pairs(g_desc.view(world * F, K, 32), g_counts.view(world * F), ai[:8], aj[:8])  # ordered after the gather
    torch.cuda.synchronize()
    comm.status()
    want = torch.cat([torch.randint(0, 256, (F, K, 32), dtype=torch.uint8, device="cuda",
                                    generator=torch.Generator(device="cuda").manual_seed(7 + r)) for r in range(world)])
    assert torch.equal(g_desc.view(world * F, K, 32), want), "gathered descriptors differ"
    assert torch.equal(after[0], m.match_pairs(want, g_counts.view(world * F), ai[:8], aj[:8])[0])
    print("rank", rank, "busy_ms %.1f host_ms %.2f" % (busy * 1e3, host * 1e3))
    assert host < 0.25 * busy, (host, busy)
elif mode == "abandon":
    if rank == 0:
        t0 = time.perf_counter()
        comm.allgather_features(desc, counts, g_desc, g_counts)
        comm.wait()
        issued = time.perf_counter() - t0
        torch.cuda.synchronize()  # bounded: the polls give up after GSLAM_HIP_COMM_TIMEOUT_S
        waited = time.perf_counter() - t0
        try:
            comm.status()
            raise SystemExit("status() did not report the abandoned exchange")
        except hip.GslamHipError as e:
            assert "gave up" in str(e), e
        try:
            comm.allgather_features(desc, counts, g_desc, g_counts)
            raise SystemExit("the next exchange did not report the failure")
        except hip.GslamHipError:
            pass
        print("rank 0 issued_ms %.2f waited_s %.2f" % (issued * 1e3, waited))
        assert issued < 0.5 and 2.0 < waited < 20.0, (issued, waited)
    else:
        time.sleep(8.0)  # never joins the exchange
comm.close(); ctx.close()
print("rank", rank, "ok")
'''


def _spawn_async(mode, name, tmp_path, extra_env=None):
    script = tmp_path / "async_worker.py"
    script.write_text(ASYNC_WORKER.format(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GSLAM_HIP_COMM_TIMEOUT_S="60")
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", mode, name], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=180)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r}:\n{outs[r][-3000:]}"
    return outs


def test_ipc_exchange_is_asynchronous_the_host_does_not_wait(tmp_path):
    """the IPC transport orders the ranks' streams through flags that bounded kernels set and poll -- an
    exchange issued behind queued GPU work returns to the host at once (as the RCCL path does), the gathered data are
    right and work enqueued after gh_comm_wait sees them."""
    outs = _spawn_async("overlap", "gslam_comm_async_%d" % os.getpid(), tmp_path)
    print("\n".join(o.strip().splitlines()[-2] for o in outs))


def test_ipc_abandoned_exchange_is_bounded_and_reported(tmp_path):
    """A peer that never joins an exchange must not hang the GPU: the polls give up after the timeout, gh_comm_status and
    the next gh_allgather* report it."""
    _spawn_async("abandon", "gslam_comm_abandon_%d" % os.getpid(), tmp_path, {"GSLAM_HIP_COMM_TIMEOUT_S": "3"})


def test_ipc_synchronous_protocol_still_agrees(tmp_path):
    """GSLAM_HIP_IPC_SYNC=1 (host barriers, kept for diagnosis) gathers the same bytes."""
    F, K, W, H = 2, 300, 640, 376
    name = "gslam_comm_sync_%d" % os.getpid()
    os.environ["GSLAM_HIP_IPC_SYNC"] = "1"
    try:
        r0, r1 = _spawn(2, F, K, W, H, name, tmp_path)
    finally:
        del os.environ["GSLAM_HIP_IPC_SYNC"]
    for k in ("g_desc", "g_counts", "g_kps", "g_idx", "g_d1", "g_d2"):
        assert r0[k].tobytes() == r1[k].tobytes(), k


def test_ipc_per_peer_streams_gather_the_same_bytes(tmp_path):
    """GSLAM_HIP_IPC_PEER_STREAMS=1: one push stream per peer with fork / join events -- what runs by itself from three ranks up
    (all xGMI links at once) -- forced on for two ranks, against the default single-stream pushes: identical gathered buffers on
    both ranks and between the two modes; also with the synchronous protocol."""
    F, K, W, H = 2, 300, 640, 376
    base = _spawn(2, F, K, W, H, "gslam_comm_ps0_%d" % os.getpid(), tmp_path)
    for extra_env in ({"GSLAM_HIP_IPC_PEER_STREAMS": "1"}, {"GSLAM_HIP_IPC_PEER_STREAMS": "1", "GSLAM_HIP_IPC_SYNC": "1"}):
        os.environ.update(extra_env)
        try:
            r0, r1 = _spawn(2, F, K, W, H, "gslam_comm_ps1_%d_%d" % (os.getpid(), len(extra_env)), tmp_path)
        finally:
            for k in extra_env:
                del os.environ[k]
        for k in ("g_desc", "g_counts", "g_kps", "g_idx", "g_d1", "g_d2"):
            assert r0[k].tobytes() == r1[k].tobytes() == base[0][k].tobytes(), k
