"""Adversarial GPU parity of the ORB front end: the HIP extractor (through the C ABI) against the CPU oracle,
bit-exact on all 28 + 32 bytes of every record, on inputs chosen to reach the rarely taken paths of
gslam_amd/csrc/orb.hip -- saturated cells (> 64 scored pixels: the list branch of fast_cells), the 32-entry cap and
the overflow slots, orb_select's overflow loads / quota cuts that split a histogram bin among ties / starved
levels / the streaming variant, 0 / 255 saturation, corners at the 19-px border -- plus a hypothesis fuzz over
(w, h, stride, K, levels, thresholds, image class, seed).  The ORB oracle is unpinned (nothing upstream to pin it
to: oracle/orb_oracle.c header), so breadth of inputs is the defence; gh_orb_plan_debug_counters proves that the
inputs reach the branches (the census is written to gpurun_out/orb_branch_census.json).

Contract: GSLAM/core/Map.h:122-195 (KeyPoint), :309-321 (N x 32 descriptor matrix); spec oracle/orb_oracle.c.
"""
import json
import os
import struct
import subprocess

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import oracle_lib
from orb_images import CLASSES

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CENSUS = {}


def _extract_gpu(ctx, frames_np, K, nlevels=8, ini_th=20, min_th=7, stride=None, census=True):
    import torch
    from gslam_amd.orb import OrbExtractor, kps_to_numpy
    B, h, w = frames_np.shape
    stride = stride or w
    buf = np.zeros((B, h, stride), np.uint8)
    buf[:, :, :w] = frames_np
    ex = OrbExtractor(ctx, w, h, max_batch=B, n_features=K, n_levels=nlevels, ini_th=ini_th, min_th=min_th)
    if census:
        ex.debug_counters(enable=True, read=False)
    d = torch.from_numpy(buf).cuda()
    kps, desc, counts = ex.extract(d)
    torch.cuda.synchronize()
    out = kps_to_numpy(kps), desc.cpu().numpy(), counts.cpu().numpy()
    dbg = ex.debug_counters(enable=False) if census else {}
    ex.close()
    return out, dbg


def _check(oracle, frames, got, K, **kw):
    kps, desc, counts = got
    for f in range(frames.shape[0]):
        ek, ed = oracle.orb_extract(frames[f], K, **kw)
        n = len(ek)
        assert counts[f] == n, f"frame {f}: count {counts[f]} vs oracle {n}"
        if kps[f, :n].tobytes() != ek.tobytes():
            bad = [i for i in range(n) if kps[f, i].tobytes() != ek[i].tobytes()]
            raise AssertionError(f"frame {f}: {len(bad)} keypoint records differ, first at {bad[0]}: "
                                 f"{kps[f, bad[0]]} vs {ek[bad[0]]}")
        assert np.array_equal(desc[f, :n], ed), f"frame {f}: descriptor bits differ"
        assert not kps[f, n:].tobytes().strip(b"\0") and not desc[f, n:].any()


def _note(name, dbg):
    acc = CENSUS.setdefault(name, {})
    for k, v in dbg.items():
        acc[k] = max(acc.get(k, 0), v) if k.startswith("max_") else acc.get(k, 0) + v


@pytest.mark.parametrize("name", sorted(CLASSES))
def test_adversarial_class_parity(ctx, oracle, name):
    for (w, h, K, stride) in [(640, 480, 1000, 640), (333, 257, 400, 336), (129, 390, 3000, 129)]:
        frames = np.stack([CLASSES[name](w, h, 1234 + 17 * i) for i in range(2)])
        got, dbg = _extract_gpu(ctx, frames, K, stride=stride)
        _check(oracle, frames, got, K)
        _note(name, dbg)


@pytest.mark.parametrize("K", [1, 7, 33, 20000])
def test_adversarial_quota_extremes(ctx, oracle, K):
    """K = 1 / 7: most levels have quota 0 (their FAST pass is skipped, the next level comes from the stand-alone
    resize); K = 20000: no level of a 640x480 frame can fill its quota except on noise."""
    for name in ("noise", "dots8", "checker2", "few_corners", "mixed"):
        frames = CLASSES[name](640, 480, 99)[None]
        got, dbg = _extract_gpu(ctx, frames, K)
        _check(oracle, frames, got, K)
        _note(f"{name}/K={K}", dbg)


def test_adversarial_thresholds_and_levels(ctx, oracle):
    for name in ("noise", "low_contrast", "binary_noise", "step_edges"):
        frames = CLASSES[name](517, 389, 7)[None]
        for nl, ini, mn, K in [(1, 20, 7, 900), (3, 254, 1, 1500), (8, 7, 7, 2000), (5, 100, 50, 600), (8, 254, 254, 500),
                               (8, 9, 8, 4000)]:
            got, dbg = _extract_gpu(ctx, frames, K, nlevels=nl, ini_th=ini, min_th=mn)
            _check(oracle, frames, got, K, nlevels=nl, ini_th=ini, min_th=mn)
            _note(f"{name}/th", dbg)


def test_adversarial_large_levels_streamed_select(ctx, oracle):
    """Levels of more than 2048 cells take orb_select's streaming variant: 2560x1440 noise (3476 cells on level 0,
    every one of them saturated) and a tie lattice."""
    for name, K in (("noise", 8000), ("dots8", 20000)):
        frames = CLASSES[name](2560, 1440, 5)[None]
        got, dbg = _extract_gpu(ctx, frames, K)
        _check(oracle, frames, got, K)
        _note(f"{name}/2560x1440", dbg)
        assert dbg["sel_streamed"] == 1


def test_adversarial_1080p_noise_batch(ctx, oracle):
    """The bench geometry (1080p, K = 2000) on saturated input, batch of 3 with a padded stride."""
    frames = np.stack([CLASSES["noise"](1920, 1080, 40 + i) for i in range(2)] + [CLASSES["mixed"](1920, 1080, 9)])
    got, dbg = _extract_gpu(ctx, frames, 2000, stride=1984)
    _check(oracle, frames, got, 2000)
    _note("noise/1080p", dbg)


def test_branch_census(ctx, oracle):
    """The inputs above must have reached every rarely taken path.  Runs its own minimal set so that it does not
    depend on test order, then merges what the other tests recorded and writes the census next to the profiles."""
    runs = {"noise": (640, 480, 1000), "dots8": (640, 480, 1000), "few_corners": (640, 480, 1000),
            "checker2": (640, 480, 1000), "binary_noise": (640, 480, 1000), "mixed": (640, 480, 5000),
            "noise/K=20000": (640, 480, 20000)}
    local = {}
    for name, (w, h, K) in runs.items():
        frames = CLASSES[name.split("/")[0]](w, h, 3)[None]
        got, dbg = _extract_gpu(ctx, frames, K)
        _check(oracle, frames, got, K)
        local[name] = dbg
        _note(name + "/census", dbg)
    n = local["noise"]
    assert n["cells"] > 0 and n["dense_cells"] > 0.9 * n["cells"], n          # nz > 64 list branch
    assert n["max_nz"] > 256 and n["max_queue"] > 1024, n
    assert n["overflow_cells"] > 0 and n["cap_cells"] > 0 and n["rank_dropped"] > 0, n
    assert n["sel_cut"] > 0, n
    # with K = 20000 the quota reaches past the 7 entries of the compact record: entries from the overflow slots are output
    assert local["noise/K=20000"]["sel_overflow_cells"] > 0, local["noise/K=20000"]
    assert n["strong_silenced"] > 0, n
    assert local["dots8"]["sel_tie_split"] > 0 or local["checker2"]["sel_tie_split"] > 0, (local["dots8"], local["checker2"])
    f = local["few_corners"]
    assert f["starved_levels"] > 0 and f["unused_slots"] > 0 and f["dense_cells"] == 0, f
    total = {}
    for name, d in CENSUS.items():
        for k, v in d.items():
            total[k] = max(total.get(k, 0), v) if k.startswith("max_") else total.get(k, 0) + v
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "orb_branch_census.json"), "w") as fp:
            json.dump({"total": total, "per_input": CENSUS}, fp, indent=1, sort_keys=True)
    except OSError:
        pass
    print("ORB branch census:", json.dumps(total, sort_keys=True))


@settings(max_examples=220, deadline=None, derandomize=True,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture, HealthCheck.data_too_large])
@given(w=st.integers(39, 210), h=st.integers(39, 170), pad=st.sampled_from([0, 0, 1, 3, 16, 61]),
       K=st.one_of(st.integers(1, 40), st.integers(41, 4000)), nlevels=st.integers(1, 8),
       min_th=st.one_of(st.integers(1, 12), st.integers(13, 254)), ini_extra=st.one_of(st.just(0), st.integers(1, 60)),
       name=st.sampled_from(sorted(CLASSES)), seed=st.integers(0, 2 ** 31 - 1), batch=st.integers(1, 3))
def test_fuzz_small_frames(ctx, oracle, w, h, pad, K, nlevels, min_th, ini_extra, name, seed, batch):
    ini_th = min(254, min_th + ini_extra)
    frames = np.stack([CLASSES[name](w, h, seed + i) for i in range(batch)])
    got, dbg = _extract_gpu(ctx, frames, K, nlevels=nlevels, ini_th=ini_th, min_th=min_th, stride=w + pad)
    _check(oracle, frames, got, K, nlevels=nlevels, ini_th=ini_th, min_th=min_th)
    _note("fuzz", dbg)


# ------------------------------------------------------------------------------------------------------------
# the same images through the FeatureDetector plugin inside a real GSLAM host process
HOST = os.path.join(ROOT, "build", "plugin_host")
LIBDIR = os.path.join(ROOT, "gslam_amd", "lib")


@pytest.mark.parametrize("name", ["noise", "binary_noise", "checker2", "dots8", "step_edges", "few_corners", "black"])
def test_adversarial_through_featuredetector_plugin(tmp_path, oracle, name):
    if not (os.path.exists(HOST) and os.path.exists(os.path.join(LIBDIR, "libgslam_featuredetector.so"))):
        pytest.skip("build/plugin_host or libgslam_featuredetector.so missing (run `make plugins` where the GSLAM headers are)")
    w, h, K = 640, 480, 1500
    img = CLASSES[name](w, h, 77)
    inp, out = tmp_path / "img.raw", tmp_path / "out.bin"
    img.tofile(inp)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = LIBDIR + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([HOST, "orb", LIBDIR, str(w), str(h), "1", str(inp), str(out), str(K)], capture_output=True,
                       text=True, timeout=300, env=env)
    ek, ed = oracle.orb_extract(img, K)
    raw = open(out, "rb").read()
    ok, n, okm, nm = struct.unpack("4i", raw[:16])
    assert ok == 1 and n == len(ek), r.stdout + r.stderr
    if n == 0:
        return
    assert r.returncode == 0, r.stdout + r.stderr
    kps = np.frombuffer(raw, oracle_lib.KP_DTYPE, n, 16)
    desc = np.frombuffer(raw, np.uint8, n * 32, 16 + n * 28).reshape(n, 32)
    assert kps.tobytes() == ek.tobytes() and np.array_equal(desc, ed)
    e = oracle.bf_match(ed, ed)
    keep = oracle.match_mask(e[0], e[1], e[2], e[0], n, 100, 0, 1, 1)
    exp = np.stack([np.nonzero(keep)[0], e[0][keep == 1]], axis=1).astype(np.int32)
    assert np.array_equal(np.frombuffer(raw, np.int32, nm * 2, 16 + n * 60).reshape(nm, 2), exp)


def test_arc_score_paths_agree_bit_for_bit(ctx, monkeypatch):
    """orb_fast_cells computes the FAST arc score on packed fp16 denormals (v_pk_minimum3_f16 / v_pk_maximum3_f16, the
    default) or with 32-bit v_min3 / v_max3 (GSLAM_HIP_ORB_PKSCORE=0, read when the plan is created): the two must give
    identical records on inputs that saturate, clip at 0 / 255 and tie (the default path is the one every other test in
    this file holds against the oracle)."""
    names = ("noise", "binary_noise", "checker1", "checker2", "step_edges", "low_contrast", "mixed")
    for name in names:
        frames = np.stack([CLASSES[name](320, 240, 99 + i) for i in range(2)])
        monkeypatch.setenv("GSLAM_HIP_ORB_PKSCORE", "1")
        (k1, d1, c1), _ = _extract_gpu(ctx, frames, 800, census=False)
        monkeypatch.setenv("GSLAM_HIP_ORB_PKSCORE", "0")
        (k0, d0, c0), _ = _extract_gpu(ctx, frames, 800, census=False)
        assert np.array_equal(c0, c1) and k0.tobytes() == k1.tobytes() and np.array_equal(d0, d1), name
    monkeypatch.delenv("GSLAM_HIP_ORB_PKSCORE")
