"""The FULL `bench.py --gpus 2` path on one GPU: two ranks launched exactly as the driver launches them
(`python -m torch.distributed.run --nproc-per-node 2 ...`), both on GPU 0, exchange through the C-ABI communicator's IPC
transport (RCCL refuses two ranks on one device), `torch.distributed` on gloo for the bootstrap / barriers
(GSLAM_BENCH_DRYRUN_BACKEND).  Checked against a world = 1 run over the same 12 global frames: the gathered descriptor
buffers, the gathered match rows and the checksum of the SHARDED all-pairs matching must be identical, every rank must be
seen through the transport, and the line must carry the N > 1 fields the driver reads."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--steps", "2", "--warmup", "1", "--width", "640", "--height", "480", "--kpts", "600", "--no-cpu-baseline", "--no-ba",
         "--no-bow", "--no-c5", "--no-host-fed", "--no-all-pairs-full", "--no-range", "--full-line"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _line(out):
    rows = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert rows, out[-3000:]
    return json.loads(rows[-1])


def test_two_rank_bench_equals_single_rank(tmp_path):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GSLAM_BENCH_DRYRUN_BACKEND="gloo", GSLAM_BENCH_VERIFY="1",
               GSLAM_HIP_COMM_TIMEOUT_S="120")
    port = _free_port()
    cmd2 = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "6"] + FLAGS
    r2 = subprocess.run(cmd2, cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
    assert r2.returncode == 0, r2.stdout[-3000:] + r2.stderr[-3000:]
    two = _line(r2.stdout)
    env1 = dict(env)
    env1.pop("GSLAM_BENCH_DRYRUN_BACKEND")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--frames", "12"] + FLAGS, cwd=tmp_path,
                        capture_output=True, text=True, timeout=600, env=env1)
    assert r1.returncode == 0, r1.stdout[-3000:] + r1.stderr[-3000:]
    one = _line(r1.stdout)
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["value"] > 0 and "gh_comm (ipc)" in two["config"]["parallelism"]
    v2, v1 = two["extra"]["verify"], one["extra"]["verify"]
    assert v2["ranks_seen"] == [0, 1] and v2["global_frames"] == v1["global_frames"] == 12
    assert v2["features_sha256"] == v1["features_sha256"], "gathered descriptors differ from the single-rank extraction"
    assert v2["matches_sha256"] == v1["matches_sha256"], "gathered match rows differ from the single-rank matching"
    assert v2["all_pairs_checksum"] == v1["all_pairs_checksum"], "sharded all-pairs matching differs"
    ap2, ap1 = two["extra"]["all_pairs_sharded"], one["extra"]["all_pairs_sharded"]
    assert ap2["frame_pairs"] == ap1["frame_pairs"] == 66 and ap2["pairs"] == ap1["pairs"] > 0
    assert "c3_stereo" in two["extra"] and "error" not in two["extra"]["c3_stereo"]  # the C3 leg ran on both ranks
    # the N > 1 line carries BOTH scaling modes (the other one as a second leg) and what the exchange moved, by which path
    ol = two["extra"]["other_scaling"]
    assert "error" not in ol and ol["scaling"] == "strong" and ol["frames_per_gpu"] == 3 and ol["global_frames"] == 6 and ol["Mkeypoints_per_s"] > 0
    xc = two["extra"]["exchange"]
    assert xc["transport"] == "gh_comm (ipc)" and "push" in xc["path"] and xc["bytes_sent_per_rank_per_step"] == 6 * 600 * 32 + 6 * 4 + 6 * 600 * 4


def test_strong_scaling_mode_splits_the_same_frames(tmp_path):
    """`--scaling strong --frames 12` on two ranks works on the SAME 12 global frames as one rank with --frames 12: identical
    gathered descriptors and match rows, the line says "strong"."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GSLAM_BENCH_DRYRUN_BACKEND="gloo", GSLAM_BENCH_VERIFY="1",
               GSLAM_HIP_COMM_TIMEOUT_S="120")
    cmd2 = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "12", "--scaling", "strong"] + FLAGS
    r2 = subprocess.run(cmd2, cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
    assert r2.returncode == 0, r2.stdout[-3000:] + r2.stderr[-3000:]
    two = _line(r2.stdout)
    env1 = dict(env)
    env1.pop("GSLAM_BENCH_DRYRUN_BACKEND")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--frames", "12", "--scaling", "strong"] + FLAGS,
                        cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env1)
    assert r1.returncode == 0, r1.stdout[-3000:] + r1.stderr[-3000:]
    one = _line(r1.stdout)
    assert two["scaling"] == one["scaling"] == "strong" and two["config"]["frames_per_gpu"] == 6 and one["config"]["frames_per_gpu"] == 12
    assert two["extra"]["other_scaling"]["scaling"] == "weak" and two["extra"]["other_scaling"]["frames_per_gpu"] == 12
    v2, v1 = two["extra"]["verify"], one["extra"]["verify"]
    assert v2["global_frames"] == v1["global_frames"] == 12
    assert v2["features_sha256"] == v1["features_sha256"] and v2["matches_sha256"] == v1["matches_sha256"]
