#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X hot path of GSLAM (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL over xGMI)

One STEP = one pass of the ORB front end + brute-force matcher over one batch of synthetic frames
already resident in HBM (BASELINE.json configs[1], "C2"): `--frames` 1920x1080 gray frames per GPU,
2000 ORB keypoints each, 256-bit BF Hamming matching of every consecutive frame pair.  With N GPUs
each rank owns `--frames` frames (weak scaling; `--scaling strong` splits them over the ranks instead: BASELINE's "1000
frames at 1/2/4/8 GPUs"), descriptors and match records are exchanged with
one RCCL all-gather each (north_star: "RCCL all-gather of descriptors/match pairs"), no other collective.

Printed JSON line (rank 0; ONE compact line <= 8 KB, `compact_line`): metric = extract+match Mkeypoints/s over the whole job, plus
  roofline      the dominant kernel's algorithmic bytes / its HIP-event time vs 8 TB/s HBM
  cpu_baseline  the CPU oracle ("port": no ORB implementation exists in the reference tree) timed on
                this box's host cores on a bounded sample of the same workload
  top level     BF Gpairs/s and BA LM-iterations/s on C4 / C5
The full record (the same keys + `extra`: every secondary leg, per-kernel times, notes) is written to
gpurun_out/bench_full.json (`full_record` in the line names it); `--full-line` prints it on stdout instead.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md)
FP64_MFMA_PEAK = 78.6e12   # FLOP/s dense f64 matrix


def orb_bytes_per_frame(w, h, k):
    """B_orb = 14.40 W H + 1021 K (8 levels, 1.2x)."""
    return 14.40 * w * h + 1021.0 * k


# SURVEY.md 8(d) split of the 14.40 W H + 1021 K figure by pipeline stage (bytes / frame)
def stage_bytes(w, h, k):
    return {
        "orb_resize": (3.018 + 2.096) * w * h,      # pyramid reads + writes
        "orb_fast_cells": 3.096 * w * h,            # FAST/NMS read of all levels
        "orb_describe": 6.191 * w * h + 1021.0 * k,  # blur read+write (fused away here) + patch, kp, desc
        "orb_select": 0.0,
    }


COMPACT_LINE_MAX = 8192    # bytes; BENCH_r05.json did not parse at 28 KB (r04's 20 KB did): VERDICT r5 item 1
_ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "algorithmic_bytes_per_launch",
                  "traffic_source")
_CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "extract_Mkpts_per_s", "match_Gpairs_per_s", "ba_iters_per_s", "cpu_model")


def compact_line(full):
    """The ONE stdout line the driver parses: the contract keys, `roofline` and `cpu_baseline` without their long notes, the in-run
    parity record and the top-level secondary rates.  Everything else (`extra`, the notes) goes to the side file `write_full`
    names.  Always <= COMPACT_LINE_MAX bytes: tests/test_bench_line.py."""
    out = {k: v for k, v in full.items() if k != "extra"}
    rf = full.get("roofline")
    if isinstance(rf, dict):
        out["roofline"] = {k: rf[k] for k in _ROOFLINE_KEYS if k in rf}
        att = rf.get("attainable")
        if isinstance(att, dict):
            out["roofline"]["lane_ops_per_pixel"] = {"min": att.get("lane_ops_per_pixel_min"), "executed": att.get("lane_ops_per_pixel_executed")}
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu_baseline"] = {k: (v[:200] if isinstance(v, str) else v) for k, v in cb.items() if k in _CPU_KEYS}
    ex = full.get("extra") or {}
    if ex.get("errors"):
        out["errors"] = {k: str(v)[:120] for k, v in ex["errors"].items()}
    out["full_record"] = full.get("full_record")
    s = json.dumps(out)
    if len(s) > COMPACT_LINE_MAX:  # cannot happen with the keys above; never let a note cost the measurement
        for k in ("errors", "parity_in_run", "full_record"):
            out.pop(k, None)
        out["config"] = {"workload": str((full.get("config") or {}).get("workload"))[:200]}
    return out


def write_full(full, world):
    """Full record (with `extra`) to gpurun_out/bench_full[_nN].json; returns the path or None (read-only tree)."""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "bench_full.json" if world == 1 else f"bench_full_n{world}.json")
        with open(path, "w") as fh:
            json.dump(full, fh, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError as exc:
        log("could not write the full record: %r" % (exc,))
        return None


def file_sha16(path):
    """first 16 hex digits of the file's sha256: the bench line names the counter file it quotes (VERDICT r4 bookkeeping (ii))"""
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def host_cores():
    """Usable host cores: affinity mask, further limited by a cgroup CPU quota if one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except Exception:
        pass
    return n


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150, help="timed steps (150 x ~20 ms = a 3 s timed region)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=1000, help="frames per GPU per step (C2: 1000); with --scaling strong: frames of the whole job")
    ap.add_argument("--no-other-scaling", action="store_true", help="N > 1: skip the second leg that runs the scaling mode not asked for")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every rank owns --frames frames (the driver's scaling bench); strong: the --frames frames of BASELINE "
                         "configs[1] are split over the ranks (1000 / N each), everything else unchanged")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--kpts", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--no-bow", action="store_true")
    ap.add_argument("--no-c3", action="store_true")
    ap.add_argument("--no-c5", action="store_true", help="skip the C5 legs (10k cams / 1M points LM, n = 60000 dense solve)")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the PCIe-inclusive leg")
    ap.add_argument("--no-range", action="store_true", help="skip the 640x480 / 3840x2160 extraction legs")
    ap.add_argument("--no-all-pairs-full", action="store_true", help="skip the 499 500-frame-pair all-pairs matching leg")
    ap.add_argument("--match-overlap", action="store_true",
                    help="N = 1: run the timed steps with the matcher on a side stream under the next step's extraction and report the "
                         "A/B against the serial schedule in extra.bf_match.overlap (measured: it hides 0.01-0.05 of 0.70 ms -- the two "
                         "kernels compete for the same CUs: profiles/BENCH_r04_n1.json).  Off by default: the extra stream would also "
                         "take a hardware queue away from the host-fed leg's copy streams")
    ap.add_argument("--torch-collectives", action="store_true",
                    help="N > 1: exchange through torch.distributed instead of the C-ABI communicator (gh_comm_*)")
    ap.add_argument("--cpu-frames", type=int, default=1000, help="bounded CPU sample (frames; 1000 = the whole C2 step, ~10 s on 16 cores)")
    ap.add_argument("--full-line", action="store_true",
                    help="print the full record (with `extra`, ~30 KB) on stdout instead of the compact line (tests, tools/show_bench.py)")
    ap.add_argument("--ba-cams", type=int, default=500)
    ap.add_argument("--ba-points", type=int, default=50000)
    ap.add_argument("--ba-iters", type=int, default=12)
    return ap.parse_args()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL's own account of the topology it found (rings / trees over xGMI, transports per peer) goes to a file per
        # rank: the evidence to read after the first real N > 1 run.  GSLAM_BENCH_NCCL_LOG=0 switches it off.
        if os.environ.get("GSLAM_BENCH_NCCL_LOG", "1") != "0" and not os.environ.get("GSLAM_BENCH_DRYRUN_BACKEND"):
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            os.environ.setdefault("NCCL_DEBUG", "INFO")
            os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH")
            os.environ.setdefault("NCCL_DEBUG_FILE", os.path.join(ROOT, "gpurun_out", "rccl_n%d_rank%s.log" % (world, os.environ.get("RANK", "0"))))
    assert torch.cuda.is_available(), "bench.py needs a GPU (gslam_amd has no CPU fallback)"
    # GSLAM_BENCH_DRYRUN_BACKEND=gloo: dry run of the multi-rank code path on a 1-GPU box (all ranks share GPU 0,
    # collectives staged through the host).  Never used for reported numbers.
    dry = os.environ.get("GSLAM_BENCH_DRYRUN_BACKEND")
    if dry:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if dry:
            dist.init_process_group(dry, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from gslam_amd import hip
    from gslam_amd.matcher import BFMatcher
    from gslam_amd.orb import OrbExtractor, synth_frames
    from gslam_amd.sharding import exchange_features_begin, exchange_matches_begin, local_pairs

    ctx = hip.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    if a.scaling == "strong":
        assert a.frames % world == 0 and a.frames // world >= 2, "--scaling strong: --frames must be a multiple of the rank count (>= 2 per rank)"
    F, W, H, K = (a.frames // world if a.scaling == "strong" else a.frames), a.width, a.height, a.kpts
    ex = OrbExtractor(ctx, W, H, max_batch=F, n_features=K)
    matcher = BFMatcher(ctx)
    # synthetic frames resident in HBM before the timed region; global frame index = rank * F + f
    frames = synth_frames(ctx, F, W, H, base_seed=0x5EED0000, first_frame=rank * F, device=dev)
    kps, desc, counts = ex.alloc_outputs(F, dev)
    # N > 1: the exchange goes through the C-ABI communicator (gh_comm_*: ncclAllGather on its own stream; what a C++
    # host would call).  If RCCL cannot be brought up through it, fall back to torch.distributed's RCCL and say so.
    comm, comm_note = None, None
    if world > 1 and not a.torch_collectives:
        from gslam_amd.sharding import Comm
        try:
            comm = Comm.ipc(ctx, rank, world, "gslam_bench_%s" % os.environ.get("MASTER_PORT", "0")) if dry else \
                Comm.rccl(ctx, rank, world)
            comm_note = "gh_comm (%s)" % comm.transport
        except Exception as exc:  # noqa: BLE001
            comm, comm_note = None, "torch.distributed (gh_comm failed: %r)" % (exc,)
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=torch.device("cpu") if dry else dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and comm is not None:  # some rank failed: everybody falls back together
            comm.close()
            comm, comm_note = None, "torch.distributed (gh_comm failed on another rank)"
    elif world > 1:
        comm_note = "torch.distributed"
    # gathered buffers (every GPU holds every frame's descriptors after the exchange)
    if comm is not None:
        g_desc_b = comm.buffer((F, K, 32), torch.uint8)
        g_counts_b = comm.buffer((F,), torch.int32)
        g_desc, g_counts = g_desc_b.view(world * F, K, 32), g_counts_b.view(world * F)
    else:
        g_desc = torch.empty((world * F, K, 32), dtype=torch.uint8, device=dev) if world > 1 else desc
        g_counts = torch.empty(world * F, dtype=torch.int32, device=dev) if world > 1 else counts
    pq, pt = local_pairs(rank, world, F, dev)
    P = pq.shape[0]
    m_full = torch.full((F, K), -1, dtype=torch.int32, device=dev)  # F rows on every rank: the last global frame has no pair
    m_idx = m_full[:P]
    m_d1 = torch.empty((P, K), dtype=torch.int16, device=dev)
    m_d2 = torch.empty((P, K), dtype=torch.int16, device=dev)
    if comm is not None:
        g_match = comm.buffer((F, K), torch.int32)
    else:
        g_match = torch.empty((world, F, K), dtype=torch.int32, device=dev) if world > 1 else None

    # Of this rank's consecutive pairs (g, g + 1) only the last one needs another rank's frame: the F - 1 purely local
    # pairs are matched straight from the local descriptors while the all-gather of the descriptors is in flight.
    n_local = min(F - 1, P)
    lq = torch.arange(0, n_local, dtype=torch.int32, device=dev)
    lt = lq + 1
    out_local = (m_idx[:n_local], m_d1[:n_local], m_d2[:n_local])
    out_rest = (m_idx[n_local:], m_d1[n_local:], m_d2[n_local:])

    match_gather = [None]  # all-gather of the previous step's match rows, still in flight during the next extraction

    def finish_match_gather():
        if comm is not None:
            comm.wait()
        elif match_gather[0] is not None:
            match_gather[0].wait()
            match_gather[0] = None

    # N = 1: the matcher of step i runs on a side stream (its own gh_ctx) while step i + 1 extracts into the other of two
    # output sets -- what a front end that streams frame batches does; with the MFMA formulation the matrix pipe does the
    # matching while the VALU does FAST.  Events order (extraction i -> match i) and (match i -> extraction i + 2).
    overlap = world == 1 and a.match_overlap
    bufs = [(kps, desc, counts)]
    step_no = [0]
    if overlap:
        bufs.append(ex.alloc_outputs(F, dev))
        side = torch.cuda.Stream(device=dev)
        ctx_side = hip.Context(local_rank, stream=side.cuda_stream)
        matcher_side = BFMatcher(ctx_side)
        ev_extracted = [torch.cuda.Event(), torch.cuda.Event()]
        ev_matched = [torch.cuda.Event(), torch.cuda.Event()]

    def step_overlapped():
        i = step_no[0] & 1
        step_no[0] += 1
        main = torch.cuda.current_stream()
        main.wait_event(ev_matched[i])  # the match that last read this output set has finished
        ex.extract(frames, bufs[i])
        ev_extracted[i].record(main)
        side.wait_event(ev_extracted[i])
        matcher_side.match_pairs(bufs[i][1], bufs[i][2], pq, pt, out=(m_idx, m_d1, m_d2))
        ev_matched[i].record(side)

    def step():
        if overlap:
            return step_overlapped()
        step_no[0] += 1
        ex.extract(frames, (kps, desc, counts))
        if comm is not None:
            comm.wait()  # the previous step's match gather (it ran behind this step's extraction)
            comm.allgather_features(desc, counts, g_desc_b, g_counts_b)
            if n_local > 0:
                matcher.match_pairs(desc, counts, lq, lt, out=out_local)
            comm.wait()
            if P > n_local:  # the boundary pair against the next rank's first frame
                matcher.match_pairs(g_desc, g_counts, pq[n_local:], pt[n_local:], out=out_rest)
            comm.allgather_matches(m_full, g_match)
        elif world > 1:
            pending = exchange_features_begin(desc, counts, g_desc, g_counts)
            finish_match_gather()  # g_match of the previous step complete before anything of this step replaces it
            if n_local > 0:
                matcher.match_pairs(desc, counts, lq, lt, out=out_local)
            pending.wait()
            if P > n_local:  # the boundary pair against the next rank's first frame
                matcher.match_pairs(g_desc, g_counts, pq[n_local:], pt[n_local:], out=out_rest)
            match_gather[0] = exchange_matches_begin(m_idx, g_match, F)
        else:
            matcher.match_pairs(g_desc, g_counts, pq, pt, out=(m_idx, m_d1, m_d2))

    def barrier():
        finish_match_gather()
        torch.cuda.synchronize()
        if comm is not None:
            comm.status()  # the asynchronous IPC transport reports an abandoned exchange only here (raises)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    ctx.prof_enable(True)  # HIP events on the launch stream, per kernel, over the timed region
    if overlap:
        ctx_side.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = ctx.prof_collect()
    ctx.prof_enable(False)
    if overlap:
        prof.update(ctx_side.prof_collect())
        ctx_side.prof_enable(False)
    last_out = bufs[(step_no[0] - 1) & 1] if overlap else bufs[0]  # the output set of the last timed step (in-run parity)
    match_overlap = None
    if overlap:
        # what a side stream would hide: the same steps with the matcher on the launch stream after the extraction and with
        # the matcher of step i on a side stream under the extraction of step i + 1
        def timed(fn, n):
            barrier()
            t1 = time.perf_counter()
            for _ in range(n):
                fn()
            barrier()
            return (time.perf_counter() - t1) * 1e3 / n

        def step_serial():
            ex.extract(frames, bufs[0])
            matcher.match_pairs(bufs[0][1], bufs[0][2], pq, pt, out=(m_idx, m_d1, m_d2))

        def step_extract_only():
            ex.extract(frames, bufs[0])

        def step_match_only():
            matcher.match_pairs(bufs[0][1], bufs[0][2], pq, pt, out=(m_idx, m_d1, m_d2))

        n_ab = max(4, min(20, a.steps))
        step_serial()
        ms_serial, ms_ext, ms_match = timed(step_serial, n_ab), timed(step_extract_only, n_ab), timed(step_match_only, n_ab)
        step_no[0] = 0
        step_overlapped()
        ms_over = timed(step_overlapped, 2 * (n_ab // 2))
        step_no[0] = 0
        match_overlap = {"serial_ms_per_step": round(ms_serial, 3), "overlapped_ms_per_step": round(ms_over, 3),
                         "extract_alone_ms": round(ms_ext, 3), "match_alone_ms": round(ms_match, 3),
                         "hidden_ms": round(ms_serial - ms_over, 3), "steps_each": n_ab,
                         "how": "matcher of step i on a side stream (second gh_ctx) under the extraction of step i + 1, two output sets"}
        step_serial()  # leave set 0 and the match rows as one complete step (in-run parity reads them)
        last_out = bufs[0]
    if world > 1 and os.environ.get("GSLAM_BENCH_VERIFY"):
        # debug aid for the dry run: the overlapped schedule must give exactly what one plain call over the gathered
        # buffers gives
        ref = matcher.match_pairs(g_desc, g_counts, pq, pt)
        torch.cuda.synchronize()
        assert torch.equal(ref[0], m_idx) and torch.equal(ref[1], m_d1) and torch.equal(ref[2], m_d2), "overlap mismatch"
        assert torch.equal(g_match[rank, :P], m_idx) and bool((g_match[rank, P:] == -1).all()), "match gather mismatch"
        assert bool((g_counts > 0).all()), "feature gather incomplete"
        log(f"rank {rank}: overlapped matching verified against the plain call ({P} pairs)")
    if world > 1:
        cdev = torch.device("cpu") if dry else dev
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        nk = counts.sum().to(torch.int64).to(cdev)
        dist.all_reduce(nk)
        total_kpts_step = int(nk.item())
    else:
        total_kpts_step = int(counts.sum().item())
    n_pairs_step = int((g_counts[pq.long()].to(torch.int64) * g_counts[pt.long()].to(torch.int64)).sum().item())

    # ---- all-pairs matching sharded over the ranks (BASELINE configs[1] read literally, at N > 1): every rank holds
    #      every frame's descriptors after the gather; the frame pairs (i < j) of the first `nf` global frames are dealt
    #      round-robin (sharding.all_pairs_block), no further exchange.  Also the cross-check quantities of
    #      tests/test_bench_multirank_gpu.py: hashes of the gathered buffers and an order-independent checksum of the
    #      all-pairs result, equal for every world size that covers the same global frames.
    verify = None
    all_pairs_multi = None
    if world > 1 or os.environ.get("GSLAM_BENCH_VERIFY"):
        import hashlib
        from gslam_amd.sharding import all_pairs_block
        finish_match_gather()
        torch.cuda.synchronize()
        nf = min(world * F, 128)
        ai, aj = all_pairs_block(rank, world, nf, dev)
        o_idx = torch.empty((max(1, ai.shape[0]), K), dtype=torch.int32, device=dev)
        o_d1 = torch.empty((max(1, ai.shape[0]), K), dtype=torch.int16, device=dev)
        o_d2 = torch.empty_like(o_d1)
        if ai.shape[0] > 0:
            matcher.match_pairs(g_desc, g_counts, ai, aj, out=(o_idx, o_d1, o_d2))
        barrier()
        t1 = time.perf_counter()
        if ai.shape[0] > 0:
            matcher.match_pairs(g_desc, g_counts, ai, aj, out=(o_idx, o_d1, o_d2))
        barrier()
        dt_ap = time.perf_counter() - t1
        w = (ai.to(torch.int64) * nf + aj.to(torch.int64) + 1)[:, None]
        chk = ((o_idx[:ai.shape[0]].to(torch.int64) + 2) * w % 1000003).sum() + (o_d1[:ai.shape[0]].to(torch.int64) & 0xFFFF).sum()
        npair = (g_counts[ai.long()].to(torch.int64) * g_counts[aj.long()].to(torch.int64)).sum()
        red = torch.stack([chk, npair]).to(torch.int64)
        if world > 1:
            cdev = torch.device("cpu") if dry else dev
            red = red.to(cdev)
            dist.all_reduce(red)
            tt = torch.tensor([dt_ap], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ap = float(tt.item())
        all_pairs_multi = {"frames": nf, "frame_pairs": nf * (nf - 1) // 2, "pairs": int(red[1].item()),
                           "Gpairs_per_s": round(int(red[1].item()) / dt_ap / 1e9, 1), "ms": round(dt_ap * 1e3, 3),
                           "how": "frame pairs dealt round-robin over %d rank(s), each on its copy of the gathered descriptors" % world}
        # every rank's id through the transport in use: the gathered buffer must read 0 .. world - 1 on every rank (the first
        # thing to look at the day this runs on N GPUs: a rank missing here means the exchange did not span the job)
        ranks_seen = [0]
        if world > 1:
            me = torch.tensor([rank], dtype=torch.int32, device=dev)
            if comm is not None:
                g_rank = comm.buffer((1,), torch.int32)
                comm.allgather(me, g_rank)
                comm.wait()
                torch.cuda.synchronize()
                ranks_seen = g_rank.view(-1).cpu().tolist()
            else:
                gl = [torch.zeros(1, dtype=torch.int32, device=torch.device("cpu") if dry else dev) for _ in range(world)]
                dist.all_gather(gl, me.to(gl[0].device))
                ranks_seen = [int(t.item()) for t in gl]
            assert ranks_seen == list(range(world)), "exchange did not reach every rank: %r" % (ranks_seen,)
        if rank == 0:
            gm = g_match if g_match is not None else m_full
            verify = {"features_sha256": hashlib.sha256(g_desc.cpu().numpy().tobytes() + g_counts.cpu().numpy().tobytes()).hexdigest(),
                      "matches_sha256": hashlib.sha256(gm.reshape(-1, K).cpu().numpy().tobytes()).hexdigest(),
                      "all_pairs_checksum": int(red[0].item()), "global_frames": world * F, "ranks_seen": ranks_seen,
                      "transport": comm_note or "none (single GPU)"}
        del o_idx, o_d1, o_d2

    # ---- N > 1: the OTHER scaling mode as a second leg, so that whichever way the driver launches this file the line holds a
    #      BASELINE-config number (strong: the 1000 frames of C2 split over the ranks) AND the per-GPU-work-fixed one (weak) --
    #      VERDICT r4 item 6.  Same kernels, same exchange, its own plan and buffers; timed like the headline (barrier, MAX over
    #      ranks), fewer steps.
    other_leg = None
    if world > 1 and not a.no_other_scaling:
        try:
            mode2 = "strong" if a.scaling == "weak" else "weak"
            F2 = a.frames // world if mode2 == "strong" else a.frames
            if mode2 == "strong" and (a.frames % world != 0 or F2 < 2):
                raise ValueError("strong leg needs --frames divisible by the rank count")
            ex2 = OrbExtractor(ctx, W, H, max_batch=F2, n_features=K)
            fr2 = synth_frames(ctx, F2, W, H, base_seed=0x5EED0000, first_frame=rank * F2, device=dev)
            o2 = ex2.alloc_outputs(F2, dev)
            pq2, pt2 = local_pairs(rank, world, F2, dev)
            P2, nl2 = pq2.shape[0], min(F2 - 1, pq2.shape[0])
            mf2 = torch.full((F2, K), -1, dtype=torch.int32, device=dev)
            md1, md2 = torch.empty((P2, K), dtype=torch.int16, device=dev), torch.empty((P2, K), dtype=torch.int16, device=dev)
            lq2 = torch.arange(0, nl2, dtype=torch.int32, device=dev)
            if comm is not None:
                gd2b, gc2b, gm2 = comm.buffer((F2, K, 32), torch.uint8), comm.buffer((F2,), torch.int32), comm.buffer((F2, K), torch.int32)
                gd2, gc2 = gd2b.view(world * F2, K, 32), gc2b.view(world * F2)
            else:
                gd2 = torch.empty((world * F2, K, 32), dtype=torch.uint8, device=dev)
                gc2 = torch.empty(world * F2, dtype=torch.int32, device=dev)
                gm2 = torch.empty((world, F2, K), dtype=torch.int32, device=dev)
            pend2 = [None]

            def step2():
                ex2.extract(fr2, o2)
                if comm is not None:
                    comm.wait()
                    comm.allgather_features(o2[1], o2[2], gd2b, gc2b)
                    if nl2 > 0:
                        matcher.match_pairs(o2[1], o2[2], lq2, lq2 + 1, out=(mf2[:nl2], md1[:nl2], md2[:nl2]))
                    comm.wait()
                    if P2 > nl2:
                        matcher.match_pairs(gd2, gc2, pq2[nl2:], pt2[nl2:], out=(mf2[nl2:P2], md1[nl2:], md2[nl2:]))
                    comm.allgather_matches(mf2, gm2)
                else:
                    pending = exchange_features_begin(o2[1], o2[2], gd2, gc2)
                    if pend2[0] is not None:
                        pend2[0].wait()
                    if nl2 > 0:
                        matcher.match_pairs(o2[1], o2[2], lq2, lq2 + 1, out=(mf2[:nl2], md1[:nl2], md2[:nl2]))
                    pending.wait()
                    if P2 > nl2:
                        matcher.match_pairs(gd2, gc2, pq2[nl2:], pt2[nl2:], out=(mf2[nl2:P2], md1[nl2:], md2[nl2:]))
                    pend2[0] = exchange_matches_begin(mf2[:P2], gm2, F2)

            def barrier2():
                if comm is not None:
                    comm.wait()
                elif pend2[0] is not None:
                    pend2[0].wait()
                    pend2[0] = None
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()

            n2 = max(2, min(a.steps, 10))
            step2()
            barrier2()
            t2 = time.perf_counter()
            for _ in range(n2):
                step2()
            barrier2()
            dt2 = torch.tensor([time.perf_counter() - t2], dtype=torch.float64, device=torch.device("cpu") if dry else dev)
            k2 = o2[2].sum().to(torch.int64).reshape(1).to(dt2.device)
            dist.all_reduce(dt2, op=dist.ReduceOp.MAX)
            dist.all_reduce(k2, op=dist.ReduceOp.SUM)
            other_leg = {"scaling": mode2, "frames_per_gpu": F2, "global_frames": world * F2, "steps": n2,
                         "ms_per_step": round(float(dt2.item()) * 1e3 / n2, 3),
                         "Mkeypoints_per_s": round(int(k2.item()) * n2 / float(dt2.item()) / 1e6, 3),
                         "workload": "C2: %d x %dx%d frames in the job, K = %d%s" % (world * F2, W, H, K, " (BASELINE's 1000 frames split over the ranks)" if mode2 == "strong" and a.frames == 1000 else "")}
            ex2.close()
            del fr2, o2, gd2, gc2, gm2
        except Exception as exc:  # noqa: BLE001  (an optional leg must never cost the headline line)
            other_leg = {"error": repr(exc)}

    # ---- C3 at N > 1 (BASELINE configs[2]: per-frame stereo extract + match sharded over the GPUs, all-gather of the
    #      records): every rank extracts its own S stereo frames, matches left-right inside the row band (needs keypoints)
    #      and left(t) -> left(t+1); the pair that crosses the rank boundary is matched from the gathered records, for which
    #      gh_allgather_features carries descriptors, counts AND keypoints.
    c3_multi = None
    if world > 1 and comm is not None and not a.no_c3:
        try:
            Ws, Hs, S = 1241, 376, 500
            ex3 = OrbExtractor(ctx, Ws, Hs, max_batch=2 * S, n_features=K)
            eyes = synth_frames(ctx, 2 * S, Ws, Hs, base_seed=0xC3000000, first_frame=rank * 2 * S, row_stride=1244, device=dev)
            o3 = ex3.alloc_outputs(2 * S, dev)
            gk3 = comm.buffer((2 * S, K, 7), torch.float32)
            gd3 = comm.buffer((2 * S, K, 32), torch.uint8)
            gc3 = comm.buffer((2 * S,), torch.int32)
            lq3 = torch.arange(0, 2 * S, 2, dtype=torch.int32, device=dev)   # L_t -> R_t (local)
            rq3 = lq3 + 1
            tq3, tt3 = lq3[:-1].contiguous(), lq3[1:].contiguous()           # L_t -> L_{t+1} inside the rank
            has_next = rank + 1 < world
            bq = torch.tensor([rank * 2 * S + 2 * S - 2], dtype=torch.int32, device=dev)  # my last left eye ...
            bt = torch.tensor([(rank + 1) * 2 * S], dtype=torch.int32, device=dev)        # ... -> the next rank's first

            def step3():
                ex3.extract(eyes, o3)
                comm.allgather_features(o3[1], o3[2], gd3, gc3, o3[0], gk3)
                matcher.match_band_pairs(o3[1], o3[0], o3[2], lq3, rq3, 2.0 / 31.0)
                matcher.match_pairs(o3[1], o3[2], tq3, tt3)
                comm.wait()
                if has_next:
                    matcher.match_pairs(gd3.view(world * 2 * S, K, 32), gc3.view(world * 2 * S), bq, bt)

            step3()
            barrier()
            t1 = time.perf_counter()
            reps3 = 3
            for _ in range(reps3):
                step3()
            barrier()
            dt3 = torch.tensor([(time.perf_counter() - t1) / reps3], dtype=torch.float64,
                               device=torch.device("cpu") if dry else dev)
            dist.all_reduce(dt3, op=dist.ReduceOp.MAX)
            k3 = o3[2].sum().to(torch.int64).to(dt3.device)
            dist.all_reduce(k3)
            c3_multi = {"workload": "C3: %d stereo frames 1241x376 x 2 eyes per GPU, K=%d per eye; band-limited L-R match, "
                                    "temporal L-L match incl. the pair across the rank boundary from gathered records "
                                    "(descriptors + counts + keypoints all-gathered)" % (S, K),
                        "stereo_frames_per_s": round(S * world / float(dt3.item()), 1),
                        "Mkeypoints_per_s": round(int(k3.item()) / float(dt3.item()) / 1e6, 2),
                        "ms_per_batch": round(float(dt3.item()) * 1e3, 3), "exchange": comm_note}
            ex3.close()
        except Exception as exc:  # noqa: BLE001  (an optional leg must never cost the headline line)
            c3_multi = {"error": repr(exc)}

    if rank != 0:
        if comm is not None:
            comm.close()
        if world > 1:
            dist.destroy_process_group()
        return

    log(f"timed region done: {elapsed:.3f}s for {a.steps} steps")
    ms_per_step = elapsed * 1e3 / a.steps
    value = total_kpts_step * a.steps / elapsed / 1e6
    # ---- roofline of the dominant kernel (largest share of HIP-event time in the timed region)
    sb = stage_bytes(W, H, K)
    orb_k = {k: v for k, v in prof.items() if k.startswith("orb_")}
    fused_pyramid = "orb_resize" not in orb_k
    if fused_pyramid:  # level l + 1 is produced inside fast_cells(l): that kernel now carries both stages' bytes
        sb["orb_fast_cells"] += sb["orb_resize"]
    dom = max(orb_k, key=lambda k: orb_k[k]["total_ms"])
    launches = prof[dom]["launches"]
    avg_ms = prof[dom]["total_ms"] / launches
    alg_bytes_per_launch = sb.get(dom, 0.0) * F * a.steps / launches
    achieved = alg_bytes_per_launch / (avg_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            rec = json.load(open(tpath)).get(dom, {})
            # measured in a separate --pmc pass at rec["frames_per_launch"] frames of the same geometry: a file collected at
            # another frame count is refused rather than rescaled (the launch shape would differ)
            traffic = int(rec["hbm_bytes_per_launch"]) if rec and int(rec.get("frames_per_launch", 0)) == F else None
        except Exception:
            traffic = None
    traffic_sha = file_sha16(tpath) if traffic is not None else None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": round(achieved * 1e9 / HBM_PEAK, 4), "traffic": traffic,
                "traffic_source": ("profiles/pmc_traffic.json sha256[:16] " + traffic_sha) if traffic_sha else None,
                "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(alg_bytes_per_launch),
                "stages": "FAST/NMS read of all levels (3.096 W H) + pyramid reads and writes (5.114 W H): the next level is "
                          "resized inside this kernel" if fused_pyramid else "FAST/NMS read of all levels (3.096 W H)",
                "note": "HBM is the contractual bound (SURVEY.md 8d); SQ counters show this kernel VALU-issue bound "
                        "(profiles/README.md), so frac understates how close the kernel is to ITS limit; its launches "
                        "share the GPU with orb_select on a side stream (1.60 ms alone, GSLAM_HIP_ORB_SELECT_OVERLAP=0)"}
    # VALU-issue view of the same kernel.  The ceiling is MEASURED in this run (gh_valu_issue_probe: register-only chains of
    # each instruction class, 8 waves / SIMD); the kernel's instruction count per launch comes from a separate
    # rocprofv3 --pmc pass (SQ_INSTS_VALU, profiles/sq_counters.json) and is labelled as such.
    probes = {}
    try:
        probes = ctx.valu_issue_probes()
    except Exception as exc:  # noqa: BLE001
        probes = {"error": repr(exc)}
    spath = os.path.join(ROOT, "profiles", "sq_counters.json")
    if os.path.exists(spath) and "fast_cells mix" in probes:
        try:
            rec = json.load(open(spath)).get(dom, {})
            if rec and int(rec.get("frames_per_launch", 0)) == F:
                insts = rec["SQ_INSTS_VALU_per_launch"]
                info0 = ctx.device_info()
                half = probes["fast_cells mix"]  # perm / pk_max / pk_min / max3 / min3 / dot4 / alignbyte: the 4.2-clock class
                full = max(v for k, v in probes.items() if isinstance(v, float))  # the fastest probed class (and / or / sub / fp32: ~2.5 clocks)
                simd_clk = info0["cu_count"] * 4 * info0["clock_khz"] * 1e3
                ach = insts / (avg_ms * 1e-3)
                roofline["valu_issue"] = {"wave_insts_per_launch": int(insts),
                                          "wave_insts_source": "profiles/sq_counters.json (SQ_INSTS_VALU, separate --pmc pass)",
                                          "wave_insts_source_sha256_16": file_sha16(spath),
                                          "achieved_Ginst_per_s": round(ach / 1e9, 1),
                                          "half_rate_class_ceiling_Ginst_per_s": round(half / 1e9, 1),
                                          "full_rate_class_ceiling_Ginst_per_s": round(full / 1e9, 1),
                                          "ceilings_source": "measured in this run (gh_valu_issue_probe, register-only chains, 8 waves per SIMD)",
                                          "clk_per_wave_inst_at_reported_clock": round(simd_clk / ach, 2),
                                          "note": "since round 4 the kernel mixes both classes (SWAR pass 1 is and / or / sub / v_bitop3): the "
                                                  "achieved rate lies between the two ceilings; SQ counters put the VALU pipe at ~92 % "
                                                  "(profiles/orb_pass1_counters_r04.txt)"}
        except Exception:
            pass
    # A second, defended ceiling for the same kernel: the lane operations the integer SPEC cannot do without,
    # per pyramid pixel, at the issue rate measured in this run -- what the kernel could reach if every instruction that is not
    # arithmetic of the spec (queueing, compaction, addressing, LDS staging) vanished.  Counted on the formulation with the fewest
    # operations known here: compass test 8 subtractions + 2 folds per 2 pixels of a 32-bit word with 16-bit fields (5.0), arc
    # score of the 11 % survivors as 16 differences + a 9-wide sliding minimum and maximum over the ring in packed 16-bit pairs
    # (40 per survivor: 4.4), 3x3 NMS of the 2.5 % scored pixels (8 compares: 0.2), the next level's bilinear resize as 2 dot2
    # + 2 mad + a quarter pack per output pixel, 1 / 1.44 output pixels per source pixel (2.95): 12.55 lane-ops per pixel.
    try:
        lane_ops_min = 5.0 + 0.11 * 40.0 + 0.025 * 8.0 + 4.25 / 1.44
        if "valu_issue" in roofline:
            vi = roofline["valu_issue"]
            px_per_launch = 3.096 * W * H * F * a.steps / launches  # pyramid pixels one launch scores (all levels: 3.096 W H per frame, 8 launches)
            wave_insts_min = lane_ops_min * px_per_launch / 64.0
            # between the two measured class ceilings, weighted like the kernel's own mix (its achieved rate over its busy share)
            rate = 0.5 * (vi["half_rate_class_ceiling_Ginst_per_s"] + vi["full_rate_class_ceiling_Ginst_per_s"]) * 1e9
            t_valu_ms = wave_insts_min / rate * 1e3
            t_hbm_ms = alg_bytes_per_launch / HBM_PEAK * 1e3
            roofline["attainable"] = {"what": "time per launch of the essential arithmetic of the integer spec at the measured issue rate "
                                              "(no queueing / addressing / staging instructions), against the HBM time of the algorithmic bytes",
                                      "lane_ops_per_pixel_min": round(lane_ops_min, 2),
                                      "lane_ops_per_pixel_executed": round(vi["wave_insts_per_launch"] * 64.0 / px_per_launch, 1),
                                      "valu_essential_ms": round(t_valu_ms, 4), "hbm_ms": round(t_hbm_ms, 4),
                                      "bound": "valu" if t_valu_ms > t_hbm_ms else "hbm",
                                      "frac": round(max(t_valu_ms, t_hbm_ms) / avg_ms, 4)}
    except Exception:  # noqa: BLE001
        pass
    orb_ms = sum(v["total_ms"] for v in orb_k.values())
    # (wall clock of the step, matcher included: the kernels overlap -- orb_select runs on a side stream -- so dividing by the SUM of
    #  kernel times understated the pipeline; VERDICT r4 bookkeeping (i))
    pipeline = {"bound": "hbm", "what": "whole step (ORB pipeline + consecutive-pair match), wall clock, vs B_orb = 14.40 W H + 1021 K",
                "achieved": round(orb_bytes_per_frame(W, H, K) * F / (ms_per_step * 1e-3) / 1e9, 1),
                "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "sum_of_orb_kernel_times_ms_per_step": round(orb_ms / a.steps, 3)}
    pipeline["frac"] = round(pipeline["achieved"] * 1e9 / HBM_PEAK, 4)
    MFMA_I8_PEAK_TOPS = 5033.0  # v_mfma_i32_16x16x64_i8 at 16 clocks per SIMD (tools/mfma_probe.hip, profiles/mfma_probe_r03.txt)
    bf_kernel = "bf_match_pairs_mfma" if "bf_match_pairs_mfma" in prof else "bf_match_pairs"
    bf_ms = prof.get(bf_kernel, {}).get("total_ms", 0.0)
    valu_ceiling = matcher.valu_probe()
    bf = {"Gpairs_per_s": round(n_pairs_step * a.steps / (bf_ms * 1e-3) / 1e9, 1) if bf_ms else None,
          "kernel": bf_kernel, "ms_per_step": round(bf_ms / a.steps, 4),
          "bound": "mfma" if bf_kernel.endswith("mfma") else "valu", "valu_ceiling_Gpairs_per_s": round(valu_ceiling / 1e9, 1),
          "pairs_per_step": n_pairs_step, "schedule": "side stream under the next extraction" if overlap else "launch stream",
          "overlap": match_overlap}
    if bf["Gpairs_per_s"]:
        # popcount kernel: against the measured xor + bcnt issue ceiling; MFMA kernel: 512 integer ops per descriptor pair
        # against the i8 MFMA peak (while it shares the chip with the next step's extraction)
        bf["frac"] = round(bf["Gpairs_per_s"] * 512e9 / (MFMA_I8_PEAK_TOPS * 1e12), 4) if bf_kernel.endswith("mfma") else \
            round(bf["Gpairs_per_s"] / bf["valu_ceiling_Gpairs_per_s"], 4)
    kernels = {k: {"launches": v["launches"], "avg_ms": round(v["total_ms"] / v["launches"], 4)}
               for k, v in prof.items()}

    # all-pairs stress of the matcher: every frame pair (i < j) among the first 128 frames
    try:
        from gslam_amd.sharding import all_pairs_block
        nf = min(F, 128)
        ai, aj = all_pairs_block(0, 1, nf, dev)
        o_idx = torch.empty((ai.shape[0], K), dtype=torch.int32, device=dev)
        o_d1 = torch.empty((ai.shape[0], K), dtype=torch.int16, device=dev)
        o_d2 = torch.empty_like(o_d1)
        matcher.match_pairs(desc, counts, ai, aj, out=(o_idx, o_d1, o_d2), mfma=False)
        torch.cuda.synchronize()
        ctx.prof_enable(True)
        matcher.match_pairs(desc, counts, ai, aj, out=(o_idx, o_d1, o_d2), mfma=False)  # the popcount kernel (north_star's formulation)
        ap = ctx.prof_collect()
        ctx.prof_enable(False)
        npairs_all = int((counts[ai.long()].to(torch.int64) * counts[aj.long()].to(torch.int64)).sum().item())
        ms_all = ap["bf_match_pairs"]["total_ms"]
        bf["all_pairs"] = {"frames": nf, "frame_pairs": int(ai.shape[0]), "pairs": npairs_all,
                           "Gpairs_per_s": round(npairs_all / (ms_all * 1e-3) / 1e9, 1),
                           "frac": round(npairs_all / (ms_all * 1e-3) / valu_ceiling, 4)}
        # the same pairs through the exact integer MFMA formulation (bf_match_mfma.hip): reported BESIDE the popcount
        # kernel, which stays the contract path; the two results must be identical
        try:
            q_idx, q_d1, q_d2 = torch.empty_like(o_idx), torch.empty_like(o_d1), torch.empty_like(o_d2)
            matcher.match_pairs(desc, counts, ai, aj, out=(q_idx, q_d1, q_d2), mfma=True)
            torch.cuda.synchronize()
            ctx.prof_enable(True)
            matcher.match_pairs(desc, counts, ai, aj, out=(q_idx, q_d1, q_d2), mfma=True)
            apm = ctx.prof_collect()
            ctx.prof_enable(False)
            ms_m = apm["bf_match_pairs_mfma"]["total_ms"]
            mfma_ops = npairs_all * 512  # algorithmic: 256 multiply-adds per descriptor pair
            bf["all_pairs_mfma"] = {
                "Gpairs_per_s": round(npairs_all / (ms_m * 1e-3) / 1e9, 1), "ms": round(ms_m, 3),
                "identical_to_popcount": bool(torch.equal(q_idx, o_idx) and torch.equal(q_d1, o_d1) and torch.equal(q_d2, o_d2)),
                "roofline": {"bound": "mfma", "unit": "TOP/s", "peak": MFMA_I8_PEAK_TOPS,
                             "peak_source": "v_mfma_i32_16x16x64_i8 at 16 clocks per SIMD (tools/mfma_probe.hip, profiles/mfma_probe_r03.txt)",
                             "achieved": round(mfma_ops / (ms_m * 1e-3) / 1e12, 1),
                             "frac": round(mfma_ops / (ms_m * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS, 4)},
                "what": "v_mfma_i32_16x16x64_i8, key = 4096 (hamming - |a| + 256) + tile straight from the accumulator, v_med3 + v_min per pair"}
            del q_idx, q_d1, q_d2
        except Exception as exc:
            bf["all_pairs_mfma"] = {"error": repr(exc)}
        del o_idx, o_d1, o_d2
        # ... and the TRUE all-pairs configuration of BASELINE configs[1]: every frame pair (i < j) of the step's F frames
        # (F = 1000: 499 500 frame pairs, 2.0e12 descriptor pairs; 8 GB of match records -- sized for the 288 GB of HBM)
        if world == 1 and not a.no_all_pairs_full and F >= 256:
            ai, aj = all_pairs_block(0, 1, F, dev)
            o_idx = torch.empty((ai.shape[0], K), dtype=torch.int32, device=dev)
            o_d1 = torch.empty((ai.shape[0], K), dtype=torch.int16, device=dev)
            o_d2 = torch.empty_like(o_d1)
            torch.cuda.synchronize()
            ctx.prof_enable(True)
            matcher.match_pairs(desc, counts, ai, aj, out=(o_idx, o_d1, o_d2))  # the product entry: dispatches to the MFMA kernel
            ap = ctx.prof_collect()
            ctx.prof_enable(False)
            c64 = counts.to(torch.int64)
            npairs_full = int((c64.sum() ** 2 - (c64 * c64).sum()).item() // 2)
            k_full = "bf_match_pairs_mfma" if "bf_match_pairs_mfma" in ap else "bf_match_pairs"
            ms_full = ap[k_full]["total_ms"]
            # size-independent check: a frame matched against itself is not in the list, so no query may be unmatched
            assert bool((o_idx[:, 0] >= 0).all()), "all-pairs: unmatched query in a non-empty pair"
            bf["all_pairs_full"] = {"frames": F, "frame_pairs": int(ai.shape[0]), "pairs": npairs_full, "kernel": k_full,
                                    "Gpairs_per_s": round(npairs_full / (ms_full * 1e-3) / 1e9, 1), "seconds": round(ms_full * 1e-3, 3),
                                    "frac": round(npairs_full * 512 / (ms_full * 1e-3) / (MFMA_I8_PEAK_TOPS * 1e12), 4)
                                    if k_full.endswith("mfma") else round(npairs_full / (ms_full * 1e-3) / valu_ceiling, 4),
                                    "match_record_GB": round(ai.shape[0] * K * 8 / 1e9, 2)}
            # the whole result again through the popcount kernel: every one of the 499 500 x K records must be identical
            p_idx, p_d1, p_d2 = torch.empty_like(o_idx), torch.empty_like(o_d1), torch.empty_like(o_d2)
            ctx.prof_enable(True)
            matcher.match_pairs(desc, counts, ai, aj, out=(p_idx, p_d1, p_d2), mfma=False)
            app = ctx.prof_collect()
            ctx.prof_enable(False)
            same = bool(torch.equal(p_idx, o_idx) and torch.equal(p_d1, o_d1) and torch.equal(p_d2, o_d2))
            ms_pop = app["bf_match_pairs"]["total_ms"]
            bf["all_pairs_full"]["popcount_kernel"] = {"seconds": round(ms_pop * 1e-3, 3),
                                                       "Gpairs_per_s": round(npairs_full / (ms_pop * 1e-3) / 1e9, 1),
                                                       "frac_of_valu_ceiling": round(npairs_full / (ms_pop * 1e-3) / valu_ceiling, 4)}
            bf["all_pairs_full"]["identical_to_popcount"] = same
            assert same, "all-pairs: the MFMA matcher and the popcount matcher disagree"
            del p_idx, p_d1, p_d2
            del o_idx, o_d1, o_d2, ai, aj
            torch.cuda.empty_cache()
    except AssertionError:
        raise  # the two matcher formulations disagree: no bench line
    except Exception as exc:
        bf["all_pairs"] = {"error": repr(exc)}

    extra = {"bf_match": bf, "kernels": kernels, "roofline_pipeline": pipeline,
             "valu_issue_probes_Ginst_per_s": {k: (round(v / 1e9, 1) if isinstance(v, float) else v) for k, v in probes.items()}}
    if other_leg is not None:
        extra["other_scaling"] = other_leg
    if world > 1:
        # what crosses the links per step and per rank, and through which path
        rccl_log = os.environ.get("NCCL_DEBUG_FILE")
        algo = None
        try:
            if rccl_log and os.path.exists(rccl_log):
                txt = open(rccl_log, errors="replace").read()
                import re as _re
                rings = len(_re.findall(r"Channel \d+/\d+ *:", txt))
                algo = {"log": os.path.relpath(rccl_log, ROOT), "channels": rings,
                        "p2p": "P2P" in txt or "via P2P" in txt, "xgmi": "XGMI" in txt.upper(), "lines": txt.count("\n")}
        except Exception:  # noqa: BLE001
            algo = None
        extra["exchange"] = {"transport": comm_note,
                             "path": ("ncclAllGather on the communicator's stream (RCCL picks ring / tree over xGMI: see rccl_log)"
                                      if comm is not None and comm.transport == "rccl" else
                                      "one device-to-device push of the rank's slice into every peer's gathered buffer (HIP IPC mappings, a "
                                      "stream per peer): the direct peer all-gather of SURVEY 8(e)" if comm is not None else "torch.distributed all_gather"),
                             "bytes_sent_per_rank_per_step": int((world - 1) * (F * K * 32 + F * 4 + F * K * 4)),
                             "bytes_received_per_rank_per_step": int((world - 1) * (F * K * 32 + F * 4 + F * K * 4)),
                             "gathers_per_step": 2, "rccl_log": algo}
    if c3_multi is not None:
        extra["c3_stereo"] = c3_multi
    if all_pairs_multi is not None:
        extra["all_pairs_sharded"] = all_pairs_multi
    if verify is not None:
        extra["verify"] = verify
    info = ctx.device_info()
    if world > 1:
        # the CPU baseline is reported at N = 1 only (contract), and the single-GPU side legs (BA, BoW) add nothing
        # to a scaling line: the other ranks have already left
        a.no_cpu_baseline = a.no_ba = a.no_bow = a.no_c3 = a.no_c5 = a.no_host_fed = a.no_all_pairs_full = a.no_range = True

    # ---- C3 (KITTI-like stereo 1241x376 x 2): extract both eyes, row-band left-right match, temporal match
    def _leg_c3():
        if a.no_c3:
            return
        Ws, Hs, S = 1241, 376, 500  # stereo frames in flight (one batch of 2 S eyes)
        stride = 1244
        ex3 = OrbExtractor(ctx, Ws, Hs, max_batch=2 * S, n_features=K)
        eyes = synth_frames(ctx, 2 * S, Ws, Hs, base_seed=0xC3000000, row_stride=stride, device=dev)  # L0 R0 L1 R1 ...
        o3 = ex3.alloc_outputs(2 * S, dev)
        lq = torch.arange(0, 2 * S, 2, dtype=torch.int32, device=dev)          # L_t -> R_t
        rq = lq + 1
        tq, tt = lq[:-1].contiguous(), lq[1:].contiguous()                     # L_t -> L_{t+1}
        band = 2.0 / 31.0

        def step3():
            ex3.extract(eyes, o3)
            lr = matcher.match_band_pairs(o3[1], o3[0], o3[2], lq, rq, band)
            tm = matcher.match_pairs(o3[1], o3[2], tq, tt)
            return lr, tm

        step3()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            step3()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / reps
        c3 = o3[2].to(torch.int64)
        pairs = int((c3[lq.long()] * c3[rq.long()]).sum().item() + (c3[tq.long()] * c3[tt.long()]).sum().item())
        c3_parity = None
        if not a.no_cpu_baseline:
            # in-run parity of the band matcher: the first stereo pair of the timed batch against the oracle
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib  # the checker
            from gslam_amd.orb import kps_to_numpy
            orc = oracle_lib.load()
            lr, _tm = step3()
            torch.cuda.synchronize()
            ek, ed, ec = orc.orb_extract_batch(eyes[:2, :, :Ws].contiguous().cpu().numpy(), K, threads=2)
            nl, nr = int(ec[0]), int(ec[1])
            e = orc.bf_match_band(ed[0, :nl], ek[0, :nl], ed[1, :nr], ek[1, :nr], band)
            ok = bool(np.array_equal(o3[2][:2].cpu().numpy(), ec)) and kps_to_numpy(o3[0][:2]).tobytes() == ek.tobytes() and \
                bool(np.array_equal(lr[0][0, :nl].cpu().numpy(), e[0])) and \
                bool(np.array_equal(lr[1][0, :nl].cpu().numpy().view(np.uint16), e[1])) and \
                bool(np.array_equal(lr[2][0, :nl].cpu().numpy().view(np.uint16), e[2]))
            c3_parity = {"stereo_pairs": 1, "left_keypoints": nl, "band_match_rows_equal": ok}
            if not ok:
                raise AssertionError("in-run parity failed: C3 band matcher")
        extra["c3_stereo"] = {"workload": "C3: %d stereo frames 1241x376 x 2 eyes, K=%d per eye, band-limited L-R match "
                                          "+ temporal L-L match" % (S, K),
                              "stereo_frames_per_s": round(S / dt, 1),
                              "Mkeypoints_per_s": round(int(c3.sum().item()) / dt / 1e6, 2),
                              "candidate_Gpairs_per_s": round(pairs / dt / 1e9, 1), "ms_per_batch": round(dt * 1e3, 3),
                              "parity_in_run": c3_parity}
        ex3.close()

    try:
        _leg_c3()
    except AssertionError:
        raise
    except Exception as exc:
        extra.setdefault("errors", {})["c3"] = repr(exc)
        log("c3 leg failed: %r" % (exc,))

    # ---- BA on this GPU, outside the timed region: LM iterations / s inside gh_ba_solve.  C4 (500 cams / 50 k points /
    #      300 k observations) and, by default at N = 1, C5 (10 k cams / 1 M points / 6 M observations: the n = 60000 dense
    #      reduced-camera solve on MFMA f64), followed by a stand-alone n = 60000 solve whose residual is asserted.
    CHOL = ("ba_potrf_flow", "ba_potf2", "ba_trsm", "ba_panel_step", "ba_syrk_panel", "ba_syrk_trailing", "ba_trsv_fwd", "ba_trsv_bwd",
            "ba_chol_fused")

    CR = ("ba_cr_factor", "ba_cr_panels", "ba_cr_update", "ba_cr_back", "ba_cr_inverse", "ba_cr_backprep")
    # the arrowhead solver (band + dense border: loop closures) adds the border's kernels and the dense corner
    ARROW = CR + ("ba_cr_border_panels", "ba_cr_border_update", "ba_cr_border_syrk", "ba_cr_border_reduce", "ba_cr_border_gather",
                  "ba_cr_border_back", "ba_cr_border_yh") + CHOL

    def ba_leg(cams, points, iters, separate_timed_run, prof_iters=None, solver="auto", graph=None):
        """solver: "auto" (band solver -- block cyclic reduction, chol_cr.hip -- when the graph is a trajectory band, what a
        deployment runs), "dense" (the MFMA factorisation BASELINE's C5 names) or "band"."""
        from gslam_amd import ba
        from gslam_amd.ba_synth import graph_census, make_graph
        name = "C5" if cams >= 10000 else "C4"
        ctx.set_ba_solver(solver)
        try:
            return _ba_leg(ba, graph_census, make_graph, name, cams, points, iters, separate_timed_run, prof_iters, graph)
        finally:
            ctx.set_ba_solver("auto")

    def _ba_leg(ba, graph_census, make_graph, name, cams, points, iters, separate_timed_run, prof_iters, graph):
        log(f"BA leg {name}: building graph")
        g = graph or make_graph(cams, points, n_obs_per_point=6, seed=1)
        census = graph_census(g)  # cameras observed, observations per camera, block fill of the reduced camera system
        log(f"BA leg {name}: warm-up solve")
        ba.solve(ctx, g, ba.default_options(max_iterations=1 if cams >= 10000 else 2))  # allocations, code load
        s = None
        if separate_timed_run:  # small graphs: time without the event overhead, then repeat with per-kernel events
            log(f"BA leg {name}: timed solve")
            for _ in range(1 if cams >= 10000 else 3):  # C4: best of 3, like the resident-graph figure below (the per-solve host work varies from run to run)
                _, _, s1, _ = ba.solve(ctx, g, ba.default_options(max_iterations=iters))
                if s is None or s1.total_ms < s.total_ms:
                    s = s1
        ctx.prof_enable(True)
        _, _, sp, _ = ba.solve(ctx, g, ba.default_options(max_iterations=prof_iters or iters))
        bprof = ctx.prof_collect()
        ctx.prof_enable(False)
        s = s or sp
        resolve = None
        if separate_timed_run and (cams < 10000 or iters >= 40):  # (C5: only with the runs to convergence)
            # the same graph kept on the device (gh_ba_graph_*): what a back end that re-optimises its window pays per solve
            G = ba.Graph(ctx, g, ba.default_options(max_iterations=iters))
            G.solve(ba.default_options(max_iterations=iters))
            ts = []
            for _ in range(3):
                G.update(cam_pose=g["cam_pose"], point_xyz=g["point_xyz"])
                t1 = time.perf_counter()
                sr, _ = G.solve(ba.default_options(max_iterations=iters))
                ts.append(time.perf_counter() - t1)
            G.close()
            assert sr.iterations == s.iterations and sr.final_cost == s.final_cost, "resident graph diverged from the one-shot solve"
            resolve = {"iters_per_s": round(sr.iterations / min(ts), 2), "ms_per_solve": round(min(ts) * 1e3, 3),
                       "what": "gh_ba_graph_solve on the resident graph after gh_ba_graph_update of poses + points "
                               "(index lists, pair lists, tables and observations stay in HBM); best of 3"}
        n = 6 * cams
        solve_flops = (n ** 3 / 3.0 + 2.0 * n * n) * sp.iterations
        chol_ms = sum(v["total_ms"] for k, v in bprof.items() if k in CHOL)
        cr_ms = sum(v["total_ms"] for k, v in bprof.items() if k in CR)
        arrow_ms = sum(v["total_ms"] for k, v in bprof.items() if k in ARROW)
        launches = sum(v["launches"] for v in bprof.values())
        used, tiles, span = ctx.last_ba_solver()
        border_cams, reordered = ctx.last_ba_order()
        border_points = ctx.last_ba_border_points()
        band_flops = None
        if used == "band" and cr_ms:
            # executed flops of the block cyclic reduction, per solve: per eliminated superblock (two neighbours) m^3 / 3 (potrf)
            # + 2 m^3 (the two panels) + 2 m^3 + 2 m^3 (two rank-m updates, one fill block) + off the critical path ~2.4 m^3
            # (L^-T, the two G products); all f64 MFMA work except the potf2 pivots
            m_sb = 64 * tiles
            band_flops = (-(-n // m_sb) - 1) * (1.0 / 3 + 6 + 2.4) * m_sb ** 3
        return {"workload": f"{name}: {cams} cams, {points} pts, {len(g['obs_cam'])} obs, Huber LM",
                "graph_census": census, "border_cams": border_cams, "border_points": border_points, "cameras_reordered_by_the_solver": reordered,
                "linear_solver": {"used": used, "camera_span": span, "half_bandwidth": 6 * span + 5,
                                  "superblock_columns": 64 * tiles if tiles else None,
                                  "solver_kernel_ms_per_iteration": round((cr_ms if used == "band" else (arrow_ms if used == "arrow" else chol_ms)) / max(1, sp.iterations), 4),
                                  "band_executed_GFLOP_per_solve": round(band_flops / 1e9, 3) if band_flops else None,
                                  "band_achieved_TFLOPs": round(band_flops * sp.iterations / (cr_ms * 1e-3) / 1e12, 3) if band_flops and cr_ms else None,
                                  "dense_GFLOP_per_solve": round((n ** 3 / 3.0 + 2.0 * n * n) / 1e9, 3),
                                  "note": ("the band solver executes %.0fx fewer flops than the dense factorisation here; it wins by that, "
                                           "not by running them faster: it is bound by launch / pivot-chain latency, not by the f64 MFMA rate"
                                           % ((n ** 3 / 3.0 + 2.0 * n * n) / band_flops)) if band_flops else None,
                                  "what": "band: block cyclic reduction over superblocks of the reduced camera system "
                                          "(gslam_amd/csrc/chol_cr.hip); arrow: the same with the cameras that loop-closure points tie "
                                          "to far-away ones as a dense border (camera_span = the band part's); dense: the MFMA f64 "
                                          "factorisation (chol.hip)"},
                "iters_per_s": round(s.iterations / (s.total_ms * 1e-3), 2), "iterations": s.iterations,
                "ms_per_iteration": round(s.total_ms / max(1, s.iterations), 3),
                "total_ms": round(s.total_ms, 2), "initial_cost": s.initial_cost, "final_cost": s.final_cost,
                "resolve": resolve, "resolve_iters_per_s": resolve["iters_per_s"] if resolve else None,
                "launches_per_iteration": round(launches / max(1, sp.iterations), 1),
                "hbm_floor": {"bytes_per_iteration": 640 * len(g["obs_cam"]) + 16 * n * n,
                              "frac": round((640 * len(g["obs_cam"]) + 16 * n * n) / HBM_PEAK / (s.total_ms * 1e-3 / max(1, s.iterations)), 4)},
                "dense_solve": ({"bound": "mfma", "n": n,
                                 "achieved_TFLOPs": round(solve_flops / (chol_ms * 1e-3) / 1e12, 3),
                                 "peak_TFLOPs": FP64_MFMA_PEAK / 1e12,
                                 "frac": round(solve_flops / (chol_ms * 1e-3) / FP64_MFMA_PEAK, 4)}
                                # (only when the dense factorisation RAN: the band / arrowhead solvers put their dense top through the same
                                #  kernels, and n^3 / 3 flops over that time is not a rate of anything -- VERDICT r5 W9 (i))
                                if chol_ms and used == "dense" else None),
                "kernels": {k: {"launches": v["launches"], "total_ms": round(v["total_ms"], 3)} for k, v in bprof.items()}}

    def shuffle_cameras(g, seed):
        """the same graph with the cameras renumbered by a random permutation: BundleGraph::keyframes is a plain vector
        (GSLAM/core/Optimizer.h:116-119), a back end fills it in co-visibility order, not in trajectory order"""
        rng = np.random.default_rng(seed)
        nc = len(g["cam_dof"])
        new_of_old = rng.permutation(nc).astype(np.int32)
        old_of_new = np.argsort(new_of_old)
        h = dict(g)
        h["cam_pose"] = np.ascontiguousarray(g["cam_pose"][old_of_new])
        h["cam_dof"] = np.ascontiguousarray(g["cam_dof"][old_of_new])
        h["obs_cam"] = new_of_old[g["obs_cam"]].astype(np.int32)
        h.pop("cam_pose_gt", None)
        return h

    def dense_solve_check(n):
        """gh_potrf_solve_dev on a well conditioned SPD system of the C5 size: TFLOP/s and ||A x - b|| / ||b|| <= 1e-10."""
        import ctypes as C
        gen = torch.Generator(device=dev).manual_seed(n)
        M = torch.randn((n, 256), dtype=torch.float64, device=dev, generator=gen)
        A = M @ M.T
        A /= 256.0
        A.diagonal().add_(4.0)
        bvec = torch.randn(n, dtype=torch.float64, device=dev, generator=gen)
        Lf, x = A.clone(), bvec.clone()
        info = C.c_int()
        ctx.prof_enable(True)
        ctx.check(hip.lib.gh_potrf_solve_dev(ctx.h, C.c_void_p(Lf.data_ptr()), n, n, C.c_void_p(x.data_ptr()), C.byref(info)))
        p = ctx.prof_collect()
        ctx.prof_enable(False)
        ms = sum(v["total_ms"] for k, v in p.items() if k in CHOL)
        res = float(torch.linalg.norm(A @ x - bvec) / torch.linalg.norm(bvec))
        assert info.value == 0 and res <= 1e-10, (info.value, res)
        fl = n ** 3 / 3.0 + 2.0 * n * n
        return {"n": n, "relative_residual": res, "ms": round(ms, 2), "achieved_TFLOPs": round(fl / (ms * 1e-3) / 1e12, 2),
                "peak_TFLOPs": FP64_MFMA_PEAK / 1e12, "frac": round(fl / (ms * 1e-3) / FP64_MFMA_PEAK, 4),
                "launches": sum(v["launches"] for v in p.values())}

    try:
        if not a.no_ba:
            from gslam_amd.ba_synth import make_graph as _mk
            g4 = _mk(a.ba_cams, a.ba_points, n_obs_per_point=6, seed=1)
            extra["ba"] = ba_leg(a.ba_cams, a.ba_points, a.ba_iters, True, graph=g4)  # auto: the band solver on this graph
            dn = ba_leg(a.ba_cams, a.ba_points, a.ba_iters, True, solver="dense", graph=g4)
            assert dn["iterations"] == extra["ba"]["iterations"] and abs(dn["final_cost"] - extra["ba"]["final_cost"]) <= 1e-9 * abs(dn["final_cost"]), \
                "the two linear solvers led the LM loop to different results"
            extra["ba"]["dense_solver"] = {k: dn[k] for k in ("iters_per_s", "ms_per_iteration", "resolve_iters_per_s", "launches_per_iteration",
                                                               "dense_solve", "linear_solver", "final_cost", "kernels")}
            # the same trajectory with loop-closure points (the band is exact by construction without them): ONE such
            # point used to send the graph to the dense solver; now the cameras it ties in form the border of an arrowhead system
            if a.ba_cams >= 200:
                nlc = 20
                g4c = _mk(a.ba_cams, a.ba_points, n_obs_per_point=6, seed=1, loop_closures=nlc)
                lc = ba_leg(a.ba_cams, a.ba_points, a.ba_iters, True, graph=g4c)
                ld = ba_leg(a.ba_cams, a.ba_points, a.ba_iters, True, solver="dense", graph=g4c)
                assert lc["iterations"] == ld["iterations"] and abs(lc["final_cost"] - ld["final_cost"]) <= 1e-9 * abs(ld["final_cost"]), \
                    "loop closures: the arrowhead and the dense solver led the LM loop to different results"
                os.environ["GSLAM_HIP_BA_POINT_BORDER"] = "0"  # the camera border of round 5 on the same graph (A/B)
                try:
                    lcc = ba_leg(a.ba_cams, a.ba_points, a.ba_iters, True, graph=g4c)
                finally:
                    del os.environ["GSLAM_HIP_BA_POINT_BORDER"]
                keys = ("iters_per_s", "ms_per_iteration", "resolve_iters_per_s", "launches_per_iteration", "linear_solver", "iterations", "final_cost")
                extra["ba"]["loop_closure"] = {"closure_points": nlc, "closure_span_cams": a.ba_cams // 2,
                                               "border_cams": lc["border_cams"], "border_points": lc["border_points"],
                                               **{k: lc[k] for k in keys}, "kernels": lc["kernels"],
                                               "dense_solver": {k: ld[k] for k in keys},
                                               "camera_border": {k: lcc[k] for k in ("iters_per_s", "resolve_iters_per_s", "border_cams", "border_points", "iterations", "final_cost")},
                                               "what": "make_graph(loop_closures=20): 20 points seen from two ends of the trajectory; "
                                                       "gh_ba_solve keeps them out of the Schur complement as the border of an arrowhead system "
                                                       "(round 6: 60 unknowns; camera_border = round 5's choice, their 51 far cameras numbered "
                                                       "last: 306 unknowns); dense_solver = the same graph through the dense factorisation (what "
                                                       "rounds 1-4 fell back to); same LM run asserted"}
            # the same graphs with the cameras in a random order (VERDICT r5 missing #2): the solver orders them for itself (ba_order.hip)
            g4s = shuffle_cameras(g4, 1)
            sh = ba_leg(a.ba_cams, a.ba_points, a.ba_iters, True, graph=g4s)
            assert sh["iterations"] == extra["ba"]["iterations"] and abs(sh["final_cost"] - extra["ba"]["final_cost"]) <= 1e-9 * abs(sh["final_cost"]), \
                "shuffled cameras: a different LM run"
            skeys = ("iters_per_s", "ms_per_iteration", "resolve_iters_per_s", "iterations", "final_cost", "border_cams", "cameras_reordered_by_the_solver")
            extra["ba"]["shuffled"] = {**{k: sh[k] for k in skeys}, "linear_solver": sh["linear_solver"]["used"], "camera_span": sh["linear_solver"]["camera_span"],
                                       "what": "the C4 graph with the cameras renumbered by a random permutation (observations in their order): one-shot rate "
                                               "incl. the ordering, resident rate without it; same LM run asserted"}
            if a.ba_cams >= 200:
                g4cs = shuffle_cameras(g4c, 2)
                shc = ba_leg(a.ba_cams, a.ba_points, a.ba_iters, True, graph=g4cs)
                assert shc["iterations"] == lc["iterations"] and abs(shc["final_cost"] - lc["final_cost"]) <= 1e-9 * abs(lc["final_cost"]), \
                    "shuffled cameras + loop closures: a different LM run"
                extra["ba"]["loop_closure"]["shuffled"] = {**{k: shc[k] for k in skeys}, "linear_solver": shc["linear_solver"]["used"]}
    except Exception as exc:  # an optional leg must never cost the headline line
        extra.setdefault("errors", {})["ba"] = repr(exc)
        log("ba leg failed: %r" % (exc,))
    try:
        if not a.no_ba and not a.no_c5 and a.ba_cams < 10000:
            from gslam_amd.ba_synth import make_graph as _mk
            g5 = _mk(10000, 1000000, n_obs_per_point=6, seed=1)
            # BASELINE configs[4] names the DENSE Schur-complement solve on MFMA: that is this leg's contract figure
            extra["ba_c5"] = ba_leg(10000, 1000000, 5, True, prof_iters=2, solver="dense", graph=g5)
            torch.cuda.empty_cache()
            # the same graph through the band solver (what gh_ba_solve picks by itself on a trajectory graph)
            bd = ba_leg(10000, 1000000, 5, True, prof_iters=2, solver="auto", graph=g5)
            assert bd["iterations"] == extra["ba_c5"]["iterations"] and \
                abs(bd["final_cost"] - extra["ba_c5"]["final_cost"]) <= 1e-9 * abs(bd["final_cost"]), "C5: the two linear solvers disagree"
            extra["ba_c5"]["band_solver"] = {k: bd[k] for k in ("iters_per_s", "ms_per_iteration", "launches_per_iteration", "linear_solver",
                                                               "final_cost", "kernels")}
            # 5 iterations are mostly set-up (index lists + 200 MB of upload) at this solver's speed: the rate of a run to convergence
            bl = ba_leg(10000, 1000000, 40, True, prof_iters=2, solver="auto", graph=g5)
            extra["ba_c5"]["band_solver"]["to_convergence"] = {k: bl[k] for k in ("iterations", "iters_per_s", "ms_per_iteration", "total_ms",
                                                                                    "initial_cost", "final_cost", "resolve_iters_per_s")}
            g5s = shuffle_cameras(g5, 1)
            sh5 = ba_leg(10000, 1000000, 40, True, prof_iters=2, solver="auto", graph=g5s)
            assert sh5["iterations"] == bl["iterations"] and abs(sh5["final_cost"] - bl["final_cost"]) <= 1e-9 * abs(bl["final_cost"]), \
                "C5 shuffled cameras: a different LM run"
            extra["ba_c5"]["shuffled"] = {**{k: sh5[k] for k in ("iterations", "iters_per_s", "ms_per_iteration", "total_ms", "final_cost",
                                                                 "resolve_iters_per_s", "border_cams", "cameras_reordered_by_the_solver")},
                                          "linear_solver": sh5["linear_solver"]["used"], "camera_span": sh5["linear_solver"]["camera_span"],
                                          "of_in_order_rate": round(sh5["iters_per_s"] / bl["iters_per_s"], 3),
                                          "what": "C5 with the cameras renumbered by a random permutation, run to convergence: the one-shot rate includes "
                                                  "the solver's own camera ordering (ba_order.hip); compare band_solver.to_convergence"}
            del g5, g5s
            torch.cuda.empty_cache()
            # C5 + 50 loop-closure points 5000 cameras apart: rounds 1-4 solved this dense (0.89 LM it/s), now band + border
            g5c = _mk(10000, 1000000, n_obs_per_point=6, seed=1, loop_closures=50, closure_span=5000)
            lc5 = ba_leg(10000, 1000000, 5, True, prof_iters=2, solver="auto", graph=g5c)
            ld5 = ba_leg(10000, 1000000, 2, False, prof_iters=2, solver="dense", graph=g5c)
            lc5_2 = ba_leg(10000, 1000000, 2, False, prof_iters=2, solver="auto", graph=g5c)
            assert lc5_2["iterations"] == ld5["iterations"] and abs(lc5_2["final_cost"] - ld5["final_cost"]) <= 1e-9 * abs(ld5["final_cost"]), \
                "C5 + loop closures: the arrowhead and the dense solver disagree"
            ll5 = ba_leg(10000, 1000000, 40, True, prof_iters=2, solver="auto", graph=g5c)
            os.environ["GSLAM_HIP_BA_POINT_BORDER"] = "0"
            try:
                ll5c = ba_leg(10000, 1000000, 40, True, prof_iters=2, solver="auto", graph=g5c)
            finally:
                del os.environ["GSLAM_HIP_BA_POINT_BORDER"]
            keys5 = ("iters_per_s", "ms_per_iteration", "launches_per_iteration", "linear_solver", "iterations", "final_cost")
            extra["ba_c5"]["loop_closure"] = {"closure_points": 50, "closure_span_cams": 5000, "border_cams": lc5["border_cams"], "border_points": lc5["border_points"],
                                              **{k: lc5[k] for k in keys5}, "kernels": lc5["kernels"],
                                              "to_convergence": {k: ll5[k] for k in ("iterations", "iters_per_s", "ms_per_iteration", "total_ms",
                                                                                     "initial_cost", "final_cost", "resolve_iters_per_s")},
                                              "dense_solver_2_iterations": {k: ld5[k] for k in ("iters_per_s", "ms_per_iteration", "iterations", "final_cost")},
                                              "camera_border_to_convergence": {k: ll5c[k] for k in ("iterations", "iters_per_s", "resolve_iters_per_s", "border_cams", "border_points", "final_cost")},
                                              "what": "make_graph(loop_closures=50, closure_span=5000); arrowhead solver (band + border); the "
                                                      "dense solver on the same graph for 2 iterations as the parity check and the rate rounds "
                                                      "1-4 had on such a graph"}
            del g5c
            torch.cuda.empty_cache()
            log("dense solve check n = 60000")
            extra["ba_c5"]["dense_solve_check"] = dense_solve_check(60000)
            torch.cuda.empty_cache()
    except Exception as exc:  # noqa: BLE001
        extra.setdefault("errors", {})["ba_c5"] = repr(exc)
        log("c5 leg failed: %r" % (exc,))

    # ---- PCIe-inclusive rate (SURVEY.md 8d defines the metric with H2D / D2H unless stated device-resident; `value` is
    #      device-resident by contract, this is the same extraction fed from pinned host memory, double-buffered)
    def _leg_host_fed():
        """gh_orb_stream_* (C ABI, no torch): a ring of pinned slots, three HIP streams (H2D DMA / extraction / pack kernel
        writing exact-size results into pinned host memory).  Frames start in pinned HOST memory, results end there."""
        import ctypes as C
        from gslam_amd.orb import OrbStream
        Fh, CH, depth = min(F, 400), 25, 3
        hp = C.c_void_p()
        ctx.check(hip.lib.gh_host_alloc_pinned(ctx.h, C.c_size_t(Fh * W * H), C.byref(hp)))
        try:
            src = frames[:Fh, :, :W].contiguous()
            torch.cuda.synchronize()
            ctx.check(hip.lib.gh_dev_download(ctx.h, hp, C.c_void_p(src.data_ptr()), C.c_size_t(Fh * W * H)))
            del src
            host = np.frombuffer((C.c_uint8 * (Fh * W * H)).from_address(hp.value), np.uint8).reshape(Fh, W * H)
            st = OrbStream(ctx, W, H, CH, depth, n_features=K)

            def run():
                tickets, total = [], 0
                for c0 in range(0, Fh - CH + 1, CH):
                    tickets.append(st.submit(host[c0:c0 + CH]))
                    if len(tickets) >= depth:
                        total += int(st.collect(tickets.pop(0), copy=False)[0][-1])
                for t in tickets:
                    total += int(st.collect(t, copy=False)[0][-1])
                return total

            run()
            t1 = time.perf_counter()
            for _ in range(3):
                kp = run()
            dt = (time.perf_counter() - t1) / 3
            nf = (Fh // CH) * CH
            extra["host_fed"] = {"what": "gh_orb_stream_* (C ABI): frames in pinned host memory -> one H2D DMA per chunk -> "
                                         "extraction -> pack kernel writes offsets + only the valid records into pinned "
                                         "host memory; 3 HIP streams, ring of %d slots x %d frames" % (depth, CH),
                                 "frames": nf, "Mkeypoints_per_s": round(kp / dt / 1e6, 2),
                                 "h2d_GB_per_s": round(nf * W * H / dt / 1e9, 1), "ms": round(dt * 1e3, 2),
                                 "us_per_frame": round(dt / nf * 1e6, 1),
                                 "link_probe": "profiles/pcie_probe_r03.txt: 57 GB/s H2D, 57 GB/s D2H, 48 GB/s each way at once"}
            st.close()
        finally:
            hip.lib.gh_host_free_pinned(ctx.h, hp)

    # ---- latency of the deployed per-frame path (GSLAM/plugins/play/main.cpp:99-155 hands frames over one at a time;
    #      evaluation/metric_time/main.cpp:4-34 measures exactly this).  (a) C ABI: one frame through gh_orb_stream
    #      (chunk 1, nothing else in flight), host clock around submit + collect; (b) through the GSLAM plugins in a C++
    #      host process (build/plugin_host lat): detectAndCompute + match + optimizePnP per frame, p50 / p99.
    def _leg_latency():
        import subprocess
        from gslam_amd.orb import OrbStream
        lat = {}
        for (w, h, k, tag) in ((640, 480, 1000, "c1_640x480_k1000"), (W, H, K, "c2_%dx%d_k%d" % (W, H, K))):
            fr = frames[0, :h, :w].contiguous().cpu().numpy().reshape(-1) if (w <= W and h <= H) else None
            if fr is None:
                continue
            st = OrbStream(ctx, w, h, 1, 1, n_features=k)
            buf = st.staging()
            host_ms, gpu_ms = [], []
            for i in range(260):
                t1 = time.perf_counter()
                buf[0, : w * h] = fr
                off, _, _, g = st.collect(st.submit(None, 1), copy=False)
                host_ms.append((time.perf_counter() - t1) * 1e3)
                gpu_ms.append(g)
            n_kp = int(off[-1])  # (off is a view of the stream's pinned block: read it before the stream goes away)
            st.close()
            host_ms, gpu_ms = np.sort(host_ms[60:]), np.sort(gpu_ms[60:])
            lat[tag] = {"extract_host_p50_ms": round(float(host_ms[len(host_ms) // 2]), 3),
                        "extract_host_p99_ms": round(float(host_ms[int(len(host_ms) * 0.99)]), 3),
                        "extract_link_to_link_p50_ms": round(float(gpu_ms[len(gpu_ms) // 2]), 3),
                        "keypoints": n_kp,
                        "what": "gh_orb_stream chunk 1: memcpy into pinned staging + H2D + extraction + packed D2H"}
        hostbin = os.path.join(ROOT, "build", "plugin_host")
        libdir = os.path.join(ROOT, "gslam_amd", "lib")
        if os.path.exists(hostbin) and os.path.exists(os.path.join(libdir, "libgslam_featuredetector.so")):
            env = dict(os.environ)
            env["LD_LIBRARY_PATH"] = libdir + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
            for (w, h, k, tag) in ((640, 480, 1000, "c1_640x480_k1000"), (W, H, K, "c2_%dx%d_k%d" % (W, H, K))):
                if w > W or h > H:
                    continue
                raw = os.path.join("/tmp", "gslam_bench_lat_%d.raw" % os.getpid())
                frames[:8, :h, :w].contiguous().cpu().numpy().tofile(raw)
                try:
                    r = subprocess.run([hostbin, "lat", libdir, str(w), str(h), "8", raw, str(k), "200"], capture_output=True,
                                       text=True, timeout=300, env=env)
                    kv = dict(tok.split("=", 1) for ln in r.stdout.splitlines() if ln.startswith("lat ")
                              for tok in ln.split()[1:])
                    if r.returncode == 0 and kv:
                        lat.setdefault(tag, {})["plugins"] = {
                            "what": "C++ GSLAM host (build/plugin_host): FeatureDetector::detectAndCompute + match against the "
                                    "previous frame + Optimizer::optimizePnP (300 points) per frame, 200 frames",
                            **{kk: round(float(kv[kk]), 3) for kk in kv if kk.endswith("_ms")},
                            "async_submit_collect_frames_per_s": round(float(kv.get("async_frames_per_s", 0.0)), 1)}
                    else:
                        lat.setdefault(tag, {})["plugins"] = {"error": (r.stdout + r.stderr)[-300:]}
                finally:
                    if os.path.exists(raw):
                        os.remove(raw)
        else:
            lat["plugins"] = "build/plugin_host missing (built only where the GSLAM headers are)"
        extra["latency"] = lat

    # ---- the ends of north_star's frame range (SURVEY.md 8d byte figures): 640x480 K=1000 (5.44 MB / frame) and
    #      3840x2160 K=2000 / K=8000 (121.5 MB / frame at K=2000), frames resident in HBM, extraction only
    def _leg_range():
        out = {}
        for (w, h, k, nfr) in ((640, 480, 1000, 2000), (3840, 2160, 2000, 100), (3840, 2160, 8000, 100)):
            exr = OrbExtractor(ctx, w, h, max_batch=nfr, n_features=k)
            fr = synth_frames(ctx, nfr, w, h, base_seed=0x5EED0000, device=dev)
            o = exr.alloc_outputs(nfr, dev)
            exr.extract(fr, o)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                exr.extract(fr, o)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / reps
            kp = int(o[2].sum().item())
            bpf = orb_bytes_per_frame(w, h, k)
            if not a.no_cpu_baseline and k <= 2000:
                # in-run parity at the ends of the frame range: the first frame of the timed batch against the oracle
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import oracle_lib  # the checker
                from gslam_amd.orb import kps_to_numpy
                ek, ed, ec = oracle_lib.load().orb_extract_batch(fr[:1, :, :w].contiguous().cpu().numpy(), k, threads=1)
                ok = int(o[2][0].item()) == int(ec[0]) and kps_to_numpy(o[0][:1]).tobytes() == ek.tobytes() and \
                    bool(np.array_equal(o[1][:1].cpu().numpy(), ed))
                extra.setdefault("parity_in_run_range", {})["%dx%d_k%d" % (w, h, k)] = {"frames": 1, "keypoints": int(ec[0]), "equal": bool(ok)}
                if not ok:
                    raise AssertionError("in-run parity failed at %dx%d" % (w, h))
            out["%dx%d_k%d" % (w, h, k)] = {"frames": nfr, "Mkeypoints_per_s": round(kp / dt / 1e6, 2),
                                            "frames_per_s": round(nfr / dt, 1), "us_per_frame": round(dt / nfr * 1e6, 2),
                                            "algorithmic_MB_per_frame": round(bpf / 1e6, 2),
                                            "achieved_GB_per_s": round(bpf * nfr / dt / 1e9, 1),
                                            "hbm_frac": round(bpf * nfr / dt / HBM_PEAK, 4), "keypoints_per_frame": kp // nfr}
            exr.close()
            del fr, o
            torch.cuda.empty_cache()
        extra["orb_range"] = out
        # the ORB-SLAM compatible mode (gh_orb_plan_set_distribution + gh_orb_plan_set_steering): per-cell FAST, quadtree,
        # continuous steering.  The compatibility path, reported beside the default mode.
        slam = {}
        for (w, h, k, nfr) in ((640, 480, 1000, 500), (1920, 1080, 2000, 100)):
            exr = OrbExtractor(ctx, w, h, max_batch=nfr, n_features=k)
            exr.set_distribution(1)
            exr.set_steering(1)
            fr = synth_frames(ctx, nfr, w, h, base_seed=0x5EED0000, device=dev)
            o = exr.alloc_outputs(nfr, dev)
            for _ in range(2):
                exr.extract(fr, o)
            torch.cuda.synchronize()
            reps, dt = 3, 1e9
            for _ in range(3):  # steady state: the best of three groups of three calls
                t1 = time.perf_counter()
                for _ in range(reps):
                    exr.extract(fr, o)
                torch.cuda.synchronize()
                dt = min(dt, (time.perf_counter() - t1) / reps)
            kp = int(o[2].sum().item())
            rec = {"frames": nfr, "Mkeypoints_per_s": round(kp / dt / 1e6, 2), "frames_per_s": round(nfr / dt, 1),
                   "timing": "best of 3 groups of 3 calls after 2 warm-up calls",
                   "us_per_frame": round(dt / nfr * 1e6, 2), "keypoints_per_frame": kp // nfr,
                   "plan_device_MB": round(exr.device_bytes() / 1e6, 1)}
            if not a.no_cpu_baseline:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import oracle_lib  # the checker
                from gslam_amd.orb import kps_to_numpy
                orc = oracle_lib.load()
                orc.orb_set_distribution(1)
                orc.orb_set_steer(1)
                try:
                    ek, ed = orc.orb_extract(fr[0, :, :w].contiguous().cpu().numpy(), k)
                finally:
                    orc.orb_set_distribution(0)
                    orc.orb_set_steer(0)
                n0 = int(o[2][0].item())
                ok = n0 == len(ek) and kps_to_numpy(o[0][:1])[0, :n0].tobytes() == ek.tobytes() and \
                    bool(np.array_equal(o[1][0, :n0].cpu().numpy(), ed))
                rec["parity_in_run"] = {"frames": 1, "keypoints": len(ek), "equal": bool(ok)}
                if not ok:
                    raise AssertionError("in-run parity of the quadtree mode failed at %dx%d" % (w, h))
            slam["%dx%d_k%d" % (w, h, k)] = rec
            exr.close()
            del fr, o
            torch.cuda.empty_cache()
        extra["orb_slam_mode"] = slam

    try:
        if not a.no_host_fed:
            _leg_host_fed()
    except Exception as exc:  # noqa: BLE001
        extra.setdefault("errors", {})["host_fed"] = repr(exc)
        log("host-fed leg failed: %r" % (exc,))
    # ---- the rest of the Optimizer boundary (round 3): pose-graph optimisation (gh_pg_solve) and the general BundleGraph
    #      with inverse-depth + XYZ landmarks and pose edges in one graph (gh_graph_solve).  LM iterations per second with
    #      the per-kernel HIP-event breakdown; parity of both is in tests/test_pg_gpu.py / test_graph_gpu.py.
    def _leg_graph():
        import numpy as np
        from gslam_amd import posegraph
        from gslam_amd.ba import default_options
        from gslam_amd.pg_synth import make_landmark_graph, make_pose_graph
        out = {}
        truth, start, dof, prob = make_pose_graph(400, 60, kind="sim3", seed=3, noise=0.01, perturb=0.05, scale_drift=0.1)
        o = default_options()
        o.max_iterations = 30
        posegraph.solve(ctx, start, dof, prob, o)

        def timed(fn, reps=5):
            """median wall time of fn() without the per-kernel events, then one profiled call for the kernel breakdown"""
            ts = []
            for _ in range(reps):
                t1 = time.perf_counter()
                r = fn()
                ts.append(time.perf_counter() - t1)
            ctx.prof_enable(True)
            fn()
            pk_ = ctx.prof_collect()
            ctx.prof_enable(False)
            return r, sorted(ts)[len(ts) // 2], pk_
        (S, sm, st), dt, pk = timed(lambda: posegraph.solve(ctx, start, dof, prob, o))
        out["pose_graph"] = {"workload": "400 SIM3 keyframes, 460 sim3 edges (n = 2800; block-sparse + dense root of 128 keyframes)",
                             "iterations": sm.iterations,
                             "iters_per_s": round(sm.iterations / dt, 1), "total_ms": round(dt * 1e3, 2), "status": int(st),
                             "cost": [sm.initial_cost, sm.final_cost],
                             "kernels_ms": {k: round(v["total_ms"], 3) for k, v in sorted(pk.items(), key=lambda kv: -kv[1]["total_ms"])[:6]}}
        # a loop-closing sized essential graph: block-sparse rounds + dense MFMA root (the dense system would be 9.8 GB)
        truth, start, dof, prob = make_pose_graph(5000, 600, kind="sim3", seed=4, noise=0.01, perturb=0.03, scale_drift=0.1)
        o = default_options()
        o.max_iterations = 15
        posegraph.solve(ctx, start, dof, prob, o)
        (S, sm, st), dt, pk = timed(lambda: posegraph.solve(ctx, start, dof, prob, o))
        sym = posegraph.bs_symbolic(5000, np.maximum(prob["sim3"][0], prob["sim3"][1]), np.minimum(prob["sim3"][0], prob["sim3"][1]))
        out["pose_graph_large"] = {"workload": "5000 SIM3 keyframes, 5600 sim3 edges (n = 35 000)", "iterations": sm.iterations,
                                   "iters_per_s": round(sm.iterations / dt, 1), "total_ms": round(dt * 1e3, 2), "status": int(st),
                                   "linear_solve_ms_total": round(sm.solve_ms_total, 2), "cost": [sm.initial_cost, sm.final_cost],
                                   "elimination": {"sparse_columns": sym["ns"], "root_keyframes": sym["nr"],
                                                   "rounds": len(sym["round_ptr"]) - 1, "blocks": len(sym["rows"]),
                                                   "block_products": sym["pair_products"]},
                                   "kernels_ms": {k: round(v["total_ms"], 3) for k, v in sorted(pk.items(), key=lambda kv: -kv[1]["total_ms"])[:8]}}
        truth, start, dof, prob = make_landmark_graph(n_frames=120, n_xyz=6000, n_idp=6000, kind="sim3", seed=5, noise=1e-3,
                                                      pose_edges=True, obs_per_point=5, outliers=0.02)
        o = default_options()
        o.huber_delta = 0.01
        o.max_iterations = 15
        posegraph.solve_graph(ctx, start, dof, prob, o)
        (S, xyz, rho, sm, st), dt, pk = timed(lambda: posegraph.solve_graph(ctx, start, dof, prob, o))
        # run-to-run reproducibility of the default mode (pre-rounded accumulation of the landmark part), asserted in the run
        S2, xyz2, rho2, sm2, st2 = posegraph.solve_graph(ctx, start, dof, prob, o)
        repro = S2.tobytes() == S.tobytes() and xyz2.tobytes() == xyz.tobytes() and \
            list(sm2.trace_cost[:sm2.trace_len]) == list(sm.trace_cost[:sm.trace_len])
        assert repro, "gh_graph_solve (deterministic = 1) differs between two runs"
        oa = default_options()
        oa.huber_delta, oa.max_iterations, oa.deterministic = 0.01, 15, 0  # plain f64 atomics: rounds 3-4, not reproducible
        posegraph.solve_graph(ctx, start, dof, prob, oa)
        (Sa, xa, ra, sma, sta), dta, pka = timed(lambda: posegraph.solve_graph(ctx, start, dof, prob, oa))
        out["general_graph"] = {"workload": "120 SIM3 keyframes + 123 pose edges + 6000 XYZ + 6000 inverse-depth landmarks, 60 000 observations",
                                "iterations": sm.iterations, "iters_per_s": round(sm.iterations / dt, 1), "total_ms": round(dt * 1e3, 2),
                                "status": int(st), "cost": [sm.initial_cost, sm.final_cost],
                                "assembly": "reproducible (gh_ba_options.deterministic = 1, default): every atomic contribution pre-rounded "
                                            "against two constants so that the sums are exact, hence order-independent",
                                "bit_identical_across_two_runs": bool(repro),
                                "atomics_mode": {"iterations": sma.iterations, "iters_per_s": round(sma.iterations / dta, 1),
                                                 "final_cost": sma.final_cost,
                                                 "what": "deterministic = 0: plain f64 atomics (the assembly of rounds 3-4)"},
                                "kernels_ms": {k: round(v["total_ms"], 3) for k, v in sorted(pk.items(), key=lambda kv: -kv[1]["total_ms"])[:8]}}
        # camera self-calibration (BundleGraph::camera + cameraDOF): the same kind of window in pixels of an OpenCV camera whose
        # focal lengths, centre and k1 k2 start 3 % off and are unknowns of the solve (a 9-row block behind the keyframes)
        from gslam_amd.pg_synth import with_camera
        truth, start, dof, base = make_landmark_graph(n_frames=120, n_xyz=10000, n_idp=2000, kind="se3", seed=6, noise=0.0,
                                                      obs_per_point=5)
        cam_true = np.array([520.0, 515.0, 318.0, 242.0, -0.28, 0.09, 1.2e-3, -8e-4, -0.01])
        cam_start = cam_true * np.array([1.03, 0.97, 1.02, 0.98, 1.03, 0.97, 1, 1, 1])
        prob = with_camera(base, cam_true, cam_start, 0b111111, pixel_noise=0.3, seed=7)
        o = default_options()
        o.huber_delta = 2.0
        o.max_iterations = 15
        posegraph.solve_graph(ctx, start, dof, prob, o)
        (S, xyz, rho, cam, sm, st), dt, pk = timed(lambda: posegraph.solve_graph(ctx, start, dof, prob, o))
        out["self_calibration"] = {"workload": "120 SE3 keyframes + 10 000 XYZ + 2000 inverse-depth landmarks, 60 000 pixel observations "
                                               "(0.3 px noise), OpenCV camera with fx fy cx cy k1 k2 free, 3 % off at the start",
                                   "iterations": sm.iterations, "iters_per_s": round(sm.iterations / dt, 1), "total_ms": round(dt * 1e3, 2),
                                   "status": int(st), "cost": [sm.initial_cost, sm.final_cost],
                                   "focal_error_rel": [float(abs(cam[0] / cam_true[0] - 1)), float(abs(cam[1] / cam_true[1] - 1))],
                                   "k1_k2_error_abs": [float(abs(cam[4] - cam_true[4])), float(abs(cam[5] - cam_true[5]))],
                                   "kernels_ms": {k: round(v["total_ms"], 3) for k, v in sorted(pk.items(), key=lambda kv: -kv[1]["total_ms"])[:8]}}
        extra["graph_solvers"] = out

    # ---- the other per-frame host entry points (pageable memory in and out, median wall time per call): RANSAC with inlier
    #      masks (Estimator boundary), triangulation, 3-D alignment (optimizeICP / fitSim3)
    def _leg_host_calls():
        from gslam_amd import estimator, posegraph
        rng = np.random.default_rng(5)
        n = 1000

        def med_us(fn, reps=30):
            for _ in range(3):
                fn()
            ts = []
            for _ in range(reps):
                t1 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t1)
            return round(sorted(ts)[len(ts) // 2] * 1e6, 1)
        src = rng.uniform(-1, 1, (n, 2))
        ph = np.c_[src, np.ones(n)] @ np.array([[1.0, 0.02, 0.1], [-0.03, 0.98, -0.05], [0.01, -0.02, 1.0]]).T
        dst = ph[:, :2] / ph[:, 2:] + rng.normal(0, 1e-3, (n, 2))
        dst[::7] += rng.uniform(-0.3, 0.3, (len(dst[::7]), 2))
        T = np.array([0, 0, 0, 1, 0.3, 0.0, 0.02])
        X = rng.uniform(-1, 1, (n, 3)) + np.array([0, 0, 4.0])
        d1 = X / np.linalg.norm(X, axis=1, keepdims=True)
        d2 = (X + T[4:]) / np.linalg.norm(X + T[4:], axis=1, keepdims=True)
        pa = rng.uniform(-2, 2, (n, 3))
        pb = 1.3 * pa[:, [1, 2, 0]] + np.array([0.5, -0.2, 1.0]) + rng.normal(0, 1e-3, (n, 3))
        extra["host_calls"] = {
            "what": "median wall time of one call, %d correspondences, pageable host arrays in and out (one pinned DMA each way "
                    "inside)" % n,
            "ransac_homography_us": med_us(lambda: estimator.estimate(ctx, estimator.HOMOGRAPHY, src, dst, 5e-3)),
            "ransac_homography_confidence_0p99_us": med_us(lambda: estimator.estimate_conf(ctx, estimator.HOMOGRAPHY, src, dst, 5e-3, 0.99)),
            "ransac_fundamental_us": med_us(lambda: estimator.estimate(ctx, estimator.FUNDAMENTAL, src, dst, 5e-3)),
            "triangulate_us": med_us(lambda: estimator.triangulate(ctx, T, d1, d2)),
            "align_sim3_with_information_us": med_us(lambda: posegraph.align_sim3(ctx, pa, pb))}

    for leg_name, leg_fn, skip in (("latency", _leg_latency, a.no_host_fed), ("orb_range", _leg_range, a.no_range),
                                   ("graph_solvers", _leg_graph, a.no_ba), ("host_calls", _leg_host_calls, a.no_host_fed)):
        try:
            if not skip:
                log("leg %s" % leg_name)
                leg_fn()
        except AssertionError:
            raise  # in-run parity
        except Exception as exc:  # noqa: BLE001
            extra.setdefault("errors", {})[leg_name] = repr(exc)
            log("%s leg failed: %r" % (leg_name, exc))

    # ---- BoW transform: the extracted descriptors of this step through GSLAM::Vocabulary-style
    #      k=10 trees (L=4 / L=6, the two sizes the reference publishes: 615.5 / 723.7 us per image on an i7-6700)
    def _leg_bow():
        if not a.no_bow:
            from gslam_amd import bow_synth
            from gslam_amd.bow import Vocabulary
            extra["bow"] = {}
            for Lv, pub in ((4, 615.5), (6, 723.7)):
                log(f"BoW leg: L={Lv}")
                voc = bow_synth.make_vocabulary(k=10, L=Lv, seed=1)
                v = Vocabulary(ctx, voc)
                nimg = min(F, 200)
                outb = v.alloc(nimg, K, dev)
                v.transform(desc[:nimg], counts[:nimg], 2, outb)
                torch.cuda.synchronize()
                ctx.prof_enable(True)
                for _ in range(3):
                    v.transform(desc[:nimg], counts[:nimg], 2, outb)
                bp = ctx.prof_collect()
                ctx.prof_enable(False)
                us_img = sum(x["total_ms"] for x in bp.values()) * 1e3 / (3 * nimg)
                one = desc[0, :int(counts[0])].cpu().numpy()
                v.transform_host(one, 2)
                t1 = time.perf_counter()
                for _ in range(20):
                    v.transform_host(one, 2)
                lat_us = (time.perf_counter() - t1) / 20 * 1e6
                rec = {"us_per_image_batched": round(us_img, 2), "us_per_image_single_host_call": round(lat_us, 1),
                       "descriptors_per_image": K, "published_us_per_image_i7_6700": pub,
                       "nodes": len(voc["nodes"]), "words_image0": int(outb[5][0])}
                if not a.no_cpu_baseline:
                    sys.path.insert(0, os.path.join(ROOT, "tests"))
                    import oracle_lib
                    orc = oracle_lib.load()
                    t1 = time.perf_counter()
                    for _ in range(5):
                        orc.bow_transform(voc, one, 2)
                    rec["cpu_oracle_us_per_image_1core"] = round((time.perf_counter() - t1) / 5 * 1e6, 1)
                    if oracle_lib.have_reference():
                        rv = oracle_lib.RefVocabulary(oracle_lib.load_reference(), bow_synth.to_gbow_bytes(voc))
                        t1 = time.perf_counter()
                        for _ in range(5):
                            rv.transform(one, 2)
                        rec["cpu_reference_us_per_image_1core"] = round((time.perf_counter() - t1) / 5 * 1e6, 1)
                        rv.close()
                extra["bow"][f"k10_L{Lv}"] = rec
                v.close()

    try:
        _leg_bow()
    except Exception as exc:  # an optional leg must never cost the headline line
        extra.setdefault("errors", {})["bow"] = repr(exc)
        log("bow leg failed: %r" % (exc,))

    # ---- CPU baseline on this box's host cores: bounded sample of the same workload (oracle = "port")
    cpu = None
    def _leg_cpu():
        nonlocal cpu
        if not a.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib  # the checker, used here only as the timed CPU leg
            oracle = oracle_lib.load()
            cores = host_cores()
            log(f"cpu baseline on {cores} cores (cpu_count={os.cpu_count()})")
            S = max(2, min(a.cpu_frames, F))
            host_frames = frames[:S, :, :W].contiguous().cpu().numpy()
            t1 = time.perf_counter()
            ckps, cdesc, ccnt = oracle.orb_extract_batch(host_frames, K, threads=cores)
            t_ext = time.perf_counter() - t1
            log(f"cpu extract done {t_ext:.2f}s")
            t1 = time.perf_counter()
            cmatch = []
            for f in range(S - 1):
                cmatch.append(oracle.bf_match(cdesc[f, :ccnt[f]], cdesc[f + 1, :ccnt[f + 1]], threads=cores))
            t_match = time.perf_counter() - t1
            # In-run parity ("bit-exact vs oracle on all frames (consecutive)"): what the oracle has
            # just computed for the timing IS the expected output of the last timed GPU step -- keypoint records, descriptor
            # bits, counts and the match rows of every consecutive pair.  A mismatch fails the run.
            from gslam_amd.orb import kps_to_numpy
            torch.cuda.synchronize()
            g_kps, g_desc_h, g_cnt = kps_to_numpy(last_out[0][:S]), last_out[1][:S].cpu().numpy(), last_out[2][:S].cpu().numpy()
            par = {"frames": S, "counts_equal": bool(np.array_equal(g_cnt, ccnt)),
                   "keypoints_equal": g_kps.tobytes() == ckps.tobytes(), "descriptors_equal": bool(np.array_equal(g_desc_h, cdesc))}
            gi, g1, g2 = m_idx[:S - 1].cpu().numpy(), m_d1[:S - 1].cpu().numpy().view(np.uint16), m_d2[:S - 1].cpu().numpy().view(np.uint16)
            rows_ok = True
            for f in range(S - 1):
                n = int(ccnt[f])
                e = cmatch[f]
                rows_ok = rows_ok and np.array_equal(gi[f, :n], e[0]) and np.array_equal(g1[f, :n], e[1]) and \
                    np.array_equal(g2[f, :n], e[2]) and bool((gi[f, n:] == -1).all())
            par["match_pairs"] = S - 1
            par["match_rows_equal"] = bool(rows_ok)
            par["matcher_kernel"] = bf.get("kernel")
            extra["parity_in_run"] = par
            if not (par["counts_equal"] and par["keypoints_equal"] and par["descriptors_equal"] and par["match_rows_equal"]):
                raise AssertionError("in-run parity failed: %r" % (par,))
            del cmatch
            # the sample has S frames and S-1 pairs; the GPU workload has F frames and F-1 pairs per rank
            cpu_kpts = int(ccnt.sum())
            cpu = {"value": round(cpu_kpts / (t_ext + t_match) / 1e6, 4), "unit": "Mkeypoints/s", "cores": cores,
                   "kind": "port",
                   "sample": f"{S} of the same {W}x{H} frames (K={K}) extracted + {S - 1} consecutive pairs matched, "
                             f"OpenMP over {cores} threads: extract {t_ext:.2f}s, match {t_match:.2f}s",
                   "extract_Mkpts_per_s": round(cpu_kpts / t_ext / 1e6, 4),
                   "match_Gpairs_per_s": round(float((ccnt[:-1].astype(np.int64) * ccnt[1:]).sum()) / t_match / 1e9, 4)}
            if oracle_lib.have_reference():
                ref = oracle_lib.load_reference()
                t1 = time.perf_counter()
                nref = min(S - 1, 8)
                for f in range(nref):
                    ref.bf_match(cdesc[f, :ccnt[f]], cdesc[f + 1, :ccnt[f + 1]], threads=cores)
                t_ref = time.perf_counter() - t1
                cpu["match_reference_kernel_Gpairs_per_s"] = round(
                    float((ccnt[:nref].astype(np.int64) * ccnt[1:nref + 1]).sum()) / t_ref / 1e9, 4)
            # single-thread figures on a small sample (SURVEY.md 8d asks for 1 thread and all cores) + the CPU itself
            S1 = min(S, 24)
            t1 = time.perf_counter()
            _, d1t, c1t = oracle.orb_extract_batch(host_frames[:S1], K, threads=1)
            t_e1 = time.perf_counter() - t1
            t1 = time.perf_counter()
            for f in range(min(S1 - 1, 6)):
                oracle.bf_match(d1t[f, :c1t[f]], d1t[f + 1, :c1t[f + 1]], threads=1)
            t_m1 = time.perf_counter() - t1
            cpu["extract_Mkpts_per_s_1thread"] = round(int(c1t.sum()) / t_e1 / 1e6, 4)
            cpu["match_Gpairs_per_s_1thread"] = round(
                float((c1t[:min(S1 - 1, 6)].astype(np.int64) * c1t[1:min(S1 - 1, 6) + 1]).sum()) / t_m1 / 1e9, 4)
            try:
                model = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")]
                cpu["cpu_model"] = model[0] if model else "unknown"
                cpu["logical_cpus"] = os.cpu_count()
            except Exception:
                pass
            log("cpu match done")
            if not a.no_ba:
                from gslam_amd.ba_synth import make_graph
                gsmall = make_graph(a.ba_cams, a.ba_points, n_obs_per_point=6, seed=1)
                t1 = time.perf_counter()
                _, _, so, _ = oracle.ba_solve(gsmall, oracle_lib.ba_options(max_iterations=2), threads=cores)
                t_ba = time.perf_counter() - t1
                cpu["ba_iters_per_s"] = round(so.iterations / t_ba, 4)
                cpu["ba_sample"] = f"2 LM iterations of the same C4 graph, {cores} threads"

    try:
        _leg_cpu()
    except AssertionError:
        raise  # in-run parity: the GPU step and the oracle disagree -- no bench line
    except Exception as exc:
        extra.setdefault("errors", {})["cpu_baseline"] = repr(exc)
        log("cpu baseline leg failed: %r" % (exc,))

    line = {
        "metric": "orb_extract_plus_bf_match_Mkeypoints_per_s", "value": round(value, 3), "unit": "Mkeypoints/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"C2: {F}x{W}x{H} frames per GPU" + (f" ({a.frames} in the job, split over the ranks)" if a.scaling == "strong" else "")
                               + f", {K} ORB kpts each, BF Hamming consecutive-pair match",
                   "frames_per_gpu": F, "width": W, "height": H, "kpts_per_frame": K,
                   "parallelism": f"frames sharded over {world} GPU(s), RCCL all-gather of descriptors + matches via {comm_note}"
                   if world > 1 else "single GPU",
                   "device": info["name"], "cu_count": info["cu_count"]},
        "roofline": roofline, "cpu_baseline": cpu, "parity_in_run": extra.get("parity_in_run"),
        # BASELINE.json's metric names three rates: the other two at top level as well (details under extra)
        "bf_match_all_pairs_Gpairs_per_s": (bf.get("all_pairs_full") or bf.get("all_pairs") or {}).get("Gpairs_per_s"),
        "bf_match_all_pairs_mfma_Gpairs_per_s": (bf.get("all_pairs_mfma") or {}).get("Gpairs_per_s"),
        "bf_match_consecutive_Gpairs_per_s": bf.get("Gpairs_per_s"),
        "ba_c4_lm_iters_per_s": (extra.get("ba") or {}).get("iters_per_s"),
        "ba_c4_linear_solver": ((extra.get("ba") or {}).get("linear_solver") or {}).get("used"),
        "ba_c4_dense_solver_lm_iters_per_s": ((extra.get("ba") or {}).get("dense_solver") or {}).get("iters_per_s"),
        "ba_c5_lm_iters_per_s": (extra.get("ba_c5") or {}).get("iters_per_s"),  # dense MFMA solve: BASELINE configs[4]
        "ba_c5_band_solver_lm_iters_per_s": ((extra.get("ba_c5") or {}).get("band_solver") or {}).get("iters_per_s"),
        "ba_c4_resident_graph_lm_iters_per_s": (extra.get("ba") or {}).get("resolve_iters_per_s"),
        "ba_c4_loop_closure_lm_iters_per_s": ((extra.get("ba") or {}).get("loop_closure") or {}).get("iters_per_s"),
        "ba_c5_loop_closure_lm_iters_per_s": ((extra.get("ba_c5") or {}).get("loop_closure") or {}).get("iters_per_s"),
        "ba_c4_shuffled_cameras_lm_iters_per_s": ((extra.get("ba") or {}).get("shuffled") or {}).get("iters_per_s"),
        "ba_c5_to_convergence_lm_iters_per_s": (((extra.get("ba_c5") or {}).get("band_solver") or {}).get("to_convergence") or {}).get("iters_per_s"),
        "ba_c5_shuffled_cameras_lm_iters_per_s": ((extra.get("ba_c5") or {}).get("shuffled") or {}).get("iters_per_s"),
        "host_fed_Mkeypoints_per_s": (extra.get("host_fed") or {}).get("Mkeypoints_per_s"),
        "extra": extra,
    }
    line["full_record"] = write_full(line, world)
    print(json.dumps(line if a.full_line else compact_line(line)), flush=True)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
