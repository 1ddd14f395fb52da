"""GPU parity of the ORB front end (HIP, through the C ABI) vs the CPU oracle: bit-exact keypoint
records (all 28 bytes) and descriptor bits, on the same synthetic frames."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu


def _extract_gpu(ctx, frames_np, K, nlevels=8, ini_th=20, min_th=7, stride=None):
    import torch
    from gslam_amd.orb import OrbExtractor, kps_to_numpy
    B, h, w = frames_np.shape
    stride = stride or w
    buf = np.zeros((B, h, stride), np.uint8)
    buf[:, :, :w] = frames_np
    ex = OrbExtractor(ctx, w, h, max_batch=B, n_features=K, n_levels=nlevels, ini_th=ini_th, min_th=min_th)
    d = torch.from_numpy(buf).cuda()
    kps, desc, counts = ex.extract(d)
    torch.cuda.synchronize()
    out = kps_to_numpy(kps), desc.cpu().numpy(), counts.cpu().numpy()
    ex.close()
    return out


def _check(oracle, frames, got, K, **kw):
    kps, desc, counts = got
    for f in range(frames.shape[0]):
        ek, ed = oracle.orb_extract(frames[f], K, **kw)
        n = len(ek)
        assert counts[f] == n, f"frame {f}: count {counts[f]} vs oracle {n}"
        assert kps[f, :n].tobytes() == ek.tobytes(), f"frame {f}: keypoint records differ"
        assert np.array_equal(desc[f, :n], ed), f"frame {f}: descriptor bits differ"
        # unused tail rows are zero-filled
        assert not kps[f, n:].tobytes().strip(b"\0") and not desc[f, n:].any()


def test_synth_frames_parity(ctx, oracle):
    import torch
    from gslam_amd.orb import synth_frames
    for (w, h, stride) in [(640, 480, 640), (333, 257, 340), (1241, 376, 1241)]:
        d = synth_frames(ctx, 3, w, h, base_seed=0x5EED0000, first_frame=5, row_stride=stride)
        torch.cuda.synchronize()
        g = d.cpu().numpy()
        for f in range(3):
            assert np.array_equal(g[f, :, :w], oracle.synth_frame(w, h, 0x5EED0000 + 5 + f))


def test_pyramid_parity(ctx, oracle):
    import torch
    from gslam_amd.orb import OrbExtractor
    g = oracle.synth_frame(640, 480, 77)
    ex = OrbExtractor(ctx, 640, 480, max_batch=1, n_features=500)
    ex.extract(torch.from_numpy(g[None]).cuda())
    torch.cuda.synchronize()
    ws, hs = oracle.orb_level_dims(640, 480)
    for l in range(1, 8):
        assert ex.level(l)[:2] == (ws[l], hs[l])
        assert np.array_equal(ex.debug_level(0, l), oracle.orb_pyramid_level(g, l)), f"level {l}"
    assert [ex.level(l)[2] for l in range(8)] == oracle.orb_quotas(500).tolist()
    ex.close()


@pytest.mark.parametrize("w,h,K", [(640, 480, 1000), (752, 480, 1500), (333, 257, 300), (1023, 577, 1200),
                                   (517, 389, 500), (2047, 129, 400), (131, 1029, 400)])
def test_extract_parity_batch(ctx, oracle, w, h, K):
    frames = np.stack([oracle.synth_frame(w, h, 0x5EED0000 + i) for i in range(4)])
    got = _extract_gpu(ctx, frames, K)
    _check(oracle, frames, got, K)


def test_extract_parity_unaligned_stride_kitti(ctx, oracle):
    """1241-wide rows (KITTI) are not dword aligned: exercises the level-0 staging copy."""
    frames = np.stack([oracle.synth_frame(1241, 376, 100 + i) for i in range(2)])
    got = _extract_gpu(ctx, frames, 2000, stride=1241)
    _check(oracle, frames, got, 2000)


def test_extract_parity_1080p(ctx, oracle):
    frames = np.stack([oracle.synth_frame(1920, 1080, 0x5EED0000 + i) for i in range(2)])
    got = _extract_gpu(ctx, frames, 2000)
    _check(oracle, frames, got, 2000)


def test_extract_parity_4k(ctx, oracle):
    """Upper end of the north-star range (3840x2160), K = 8000: > 7900 cells on level 0, quota 1737."""
    frames = oracle.synth_frame(3840, 2160, 0x4B000000)[None]
    got = _extract_gpu(ctx, frames, 8000)
    _check(oracle, frames, got, 8000)
    assert got[2][0] == 8000


def test_extract_parity_levels_and_thresholds(ctx, oracle):
    frames = np.stack([oracle.synth_frame(640, 480, 500 + i) for i in range(2)])
    for nl, ini, mn, K in [(1, 20, 7, 800), (4, 30, 10, 1200), (8, 12, 5, 5000), (8, 20, 7, 7)]:
        got = _extract_gpu(ctx, frames, K, nlevels=nl, ini_th=ini, min_th=mn)
        _check(oracle, frames, got, K, nlevels=nl, ini_th=ini, min_th=mn)


def test_sparse_and_flat_images(ctx, oracle):
    """Few textured cells -> levels under quota; flat image -> zero keypoints."""
    g = np.full((480, 640), 90, np.uint8)
    g[100:228, 200:328] = oracle.synth_frame(128, 128, 3)
    frames = np.stack([g, np.full((480, 640), 90, np.uint8)])
    got = _extract_gpu(ctx, frames, 1000)
    _check(oracle, frames, got, 1000)
    assert got[2][1] == 0 and 0 < got[2][0] < 1000


def test_extract_host_entry_point(ctx, oracle):
    from gslam_amd.orb import OrbExtractor
    g = oracle.synth_frame(640, 480, 31337)
    ex = OrbExtractor(ctx, 640, 480, max_batch=1, n_features=1000)
    kps, desc = ex.extract_host(g)
    ek, ed = oracle.orb_extract(g, 1000)
    assert kps.tobytes() == ek.tobytes() and np.array_equal(desc, ed)
    ex.close()


def test_batch_slot_independence(ctx, oracle):
    """The same frame in different batch slots (and batch sizes) gives identical records."""
    import torch
    from gslam_amd.orb import OrbExtractor, synth_frames
    ex = OrbExtractor(ctx, 640, 480, max_batch=16, n_features=1000)
    fr = synth_frames(ctx, 16, 640, 480, base_seed=1000)
    k16, d16, c16 = [t.clone() for t in ex.extract(fr)]
    k1, d1, c1 = ex.extract(fr[5:6].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(k16[5].view(torch.int32), k1[0].view(torch.int32)) and torch.equal(d16[5], d1[0])
    rev = torch.flip(fr, dims=[0]).contiguous()
    kr, dr, cr = ex.extract(rev)
    torch.cuda.synchronize()
    assert torch.equal(torch.flip(kr, dims=[0]).view(torch.int32), k16.view(torch.int32))
    assert torch.equal(torch.flip(dr, dims=[0]), d16) and torch.equal(torch.flip(cr, dims=[0]), c16)
    ex.close()


def test_bgr_to_gray_parity(ctx, oracle):
    import torch
    from gslam_amd import hip
    rng = np.random.default_rng(8)
    for ch in (3, 4):
        bgr = rng.integers(0, 256, (45, 67, ch), dtype=np.uint8)
        d = torch.from_numpy(bgr).cuda()
        out = torch.empty((45, 67), dtype=torch.uint8, device="cuda")
        ctx.check(hip.lib.gh_bgr_to_gray_dev(ctx.h, C.c_void_p(d.data_ptr()), 67, 45, ch, 67 * ch,
                                             C.c_void_p(out.data_ptr()), 67))
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), oracle.bgr_to_gray(bgr))


def test_extract_full_size_properties(ctx, oracle):
    """BASELINE configs[1] geometry (1920x1080, K=2000) at a batch the oracle cannot cover in seconds: size-independent
    properties instead -- batch-split invariance (one batch of 192 == three batches of 64, bitwise), record invariants,
    and an oracle spot check on three frames spread over the batch."""
    import torch
    from gslam_amd.orb import OrbExtractor, kps_to_numpy, synth_frames
    B, W, H, K = 192, 1920, 1080, 2000
    ex = OrbExtractor(ctx, W, H, max_batch=B, n_features=K)
    fr = synth_frames(ctx, B, W, H, base_seed=0x5EED0000)
    k, d, c = [t.clone() for t in ex.extract(fr)]
    for s in range(0, B, 64):
        k2, d2, c2 = ex.extract(fr[s:s + 64])
        torch.cuda.synchronize()
        assert torch.equal(k2.view(torch.int32), k[s:s + 64].view(torch.int32))
        assert torch.equal(d2, d[s:s + 64]) and torch.equal(c2, c[s:s + 64])
    kp, dn, cn = kps_to_numpy(k), d.cpu().numpy(), c.cpu().numpy()
    assert cn.min() > 0 and cn.max() <= K
    for f in range(B):
        n = int(cn[f])
        rec = kp[f, :n]
        assert rec["octave"].min() >= 0 and rec["octave"].max() <= 7
        assert (np.diff(rec["octave"]) >= 0).all(), "keypoints are grouped by level"
        assert rec["response"].min() > 7 and rec["response"].max() <= 255
        assert (rec["x"] >= 19).all() and (rec["x"] < W - 19 + 1).all() and (rec["y"] >= 19).all() and (rec["y"] < H - 19 + 1).all()
        assert set(np.unique(rec["angle"])) <= set(12.0 * np.arange(30))
        assert not kp[f, n:].tobytes().strip(b"\0") and not dn[f, n:].any()
    frames_np = fr[[0, 95, 191]].cpu().numpy()
    for i, f in enumerate((0, 95, 191)):
        ek, ed = oracle.orb_extract(frames_np[i], K)
        assert cn[f] == len(ek) and kp[f, :len(ek)].tobytes() == ek.tobytes() and np.array_equal(dn[f, :len(ek)], ed)
    ex.close()


def test_runtime_pattern_injection(ctx, oracle):
    """gh_orb_plan_set_pattern: (1) the built-in unrotated pattern reproduces the default bits, (2) a different pattern
    matches the oracle with the same pattern installed, (3) a pattern that leaves the blurred patch is rejected."""
    import os
    import sys
    import torch
    from gslam_amd import hip
    from gslam_amd.orb import OrbExtractor, kps_to_numpy, synth_frames
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_orb_tables as T
    base = np.array(T.gen_pattern(), np.int8)
    ex = OrbExtractor(ctx, 640, 480, max_batch=2, n_features=700)
    frames = synth_frames(ctx, 2, 640, 480, base_seed=0x5EED0100)
    k0, d0, c0 = [t.clone() for t in ex.extract(frames)]
    ex.set_pattern(base)
    k1, d1, c1 = ex.extract(frames)
    torch.cuda.synchronize()
    assert torch.equal(d0, d1) and torch.equal(c0, c1) and torch.equal(k0.view(torch.int32), k1.view(torch.int32))
    rng = np.random.default_rng(5)
    pat = rng.integers(-9, 10, (256, 4)).astype(np.int8)
    pat[pat[:, 0] == pat[:, 2], 2] += 1
    ex.set_pattern(pat)
    k2, d2, c2 = ex.extract(frames)
    torch.cuda.synchronize()
    assert oracle.orb_set_pattern(pat)
    try:
        host = frames.cpu().numpy()
        for f in range(2):
            ek, ed = oracle.orb_extract(host[f], 700)
            n = int(c2[f])
            assert n == len(ek) and kps_to_numpy(k2)[f, :n].tobytes() == ek.tobytes()
            assert np.array_equal(d2[f, :n].cpu().numpy(), ed)
    finally:
        oracle.orb_set_pattern(None)
    assert not torch.equal(d2, d0)
    bad = pat.copy()
    bad[3] = [13, 13, 0, 0]
    with pytest.raises(hip.GslamHipError):
        ex.set_pattern(bad)
    ex.close()


def test_roi_view_whose_allocation_ends_at_the_last_pixel(ctx, oracle):
    """a single frame with row_stride > width whose buffer ends at (h-1) * stride + w (an ROI view of a larger
    image) must not be read past its end.  Device path: frame_stride = 0 with batch = 1 stages the frame row by row; host
    path: gh_orb_extract_host copies exactly (h-1) * stride + w bytes.  Results equal the oracle on the dense image."""
    import ctypes as C
    import torch
    from gslam_amd import hip
    from gslam_amd.orb import KP_DTYPE, OrbExtractor, kps_to_numpy
    w, h, stride, K = 400, 300, 448, 500
    img = oracle.synth_frame(w, h, 0x5EED0200)
    ek, ed = oracle.orb_extract(img, K)
    n_bytes = (h - 1) * stride + w
    flat = np.zeros(n_bytes, np.uint8)
    for y in range(h):
        flat[y * stride:y * stride + w] = img[y]
    ex = OrbExtractor(ctx, w, h, max_batch=1, n_features=K)
    # device: the allocation is exactly n_bytes long (16-byte aligned pointer and stride, so the zero-copy path would apply)
    dbuf = torch.from_numpy(flat).cuda()
    kps, desc, counts = ex.alloc_outputs(1)
    ctx.check(hip.lib.gh_orb_extract_dev(ex.plan, C.c_void_p(dbuf.data_ptr()), 1, C.c_size_t(0), stride,
                                         C.c_void_p(kps.data_ptr()), C.c_void_p(desc.data_ptr()), C.c_void_p(counts.data_ptr())))
    torch.cuda.synchronize()
    n = int(counts[0])
    assert n == len(ek) and kps_to_numpy(kps)[0, :n].tobytes() == ek.tobytes() and np.array_equal(desc[0, :n].cpu().numpy(), ed)
    # host: a buffer that really ends at the last pixel (guard page semantics are the OS's; here: exact-size numpy array)
    hk = np.zeros(K, KP_DTYPE)
    hd = np.zeros((K, 32), np.uint8)
    hn = C.c_int32()
    ctx.check(hip.lib.gh_orb_extract_host(ex.plan, flat.ctypes.data_as(C.c_void_p), stride, hk.ctypes.data_as(C.c_void_p),
                                          hd.ctypes.data_as(C.c_void_p), C.byref(hn)))
    assert hn.value == len(ek) and hk[:hn.value].tobytes() == ek.tobytes() and np.array_equal(hd[:hn.value], ed)
    ex.close()


def test_small_call_graph_cache_survives_fresh_and_rotating_buffers(ctx):
    """The hipGraph replay of small calls (gh_orb_extract_dev): (a) more rotating output sets than the cache holds -- evicted
    graphs are retired, not destroyed under a queued launch; (b) fresh output buffers every call -- the plan gives up
    capturing; the records stay identical throughout."""
    import torch
    from gslam_amd.orb import OrbExtractor, synth_frames
    for fresh_first in (False, True):
        ex = OrbExtractor(ctx, 320, 240, max_batch=1, n_features=300)
        fr = synth_frames(ctx, 1, 320, 240, base_seed=0x5EED0000)
        def same(out, ref):  # (KeyPoint.class_id = -1 reads as a NaN through the float32 view: compare the bits)
            return all(torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a,
                                   b.view(torch.int32) if b.dtype == torch.float32 else b) for a, b in zip(out, ref))

        ref = [t.clone() for t in ex.extract(fr)]
        torch.cuda.synchronize()
        assert int(ref[2][0]) > 100
        if fresh_first:
            for it in range(24):  # fresh buffers every call
                out = ex.extract(fr)
                torch.cuda.synchronize()
                assert same(out, ref), it
        sets = [ex.alloc_outputs(1) for _ in range(20)]  # more than the 16-entry ring
        for rnd in range(3):
            for o in sets:
                ex.extract(fr, o)
            torch.cuda.synchronize()
            for o in sets:
                assert same(o, ref), (fresh_first, rnd)
        ex.close()


def _steer_pattern():
    from test_orb_oracle import _canonical_like_pattern
    return _canonical_like_pattern()


@pytest.mark.parametrize("w,h,K,B,pattern", [(640, 480, 1000, 2, False), (640, 480, 1000, 2, True), (1920, 1080, 2000, 2, True),
                                              (161, 123, 200, 3, True), (1241, 376, 1500, 2, False)])
def test_continuous_steering_parity(ctx, oracle, w, h, K, B, pattern):
    """gh_orb_plan_set_steering(plan, 1): fastAtan2 orientation + per-keypoint rotation of the test pattern (the
    OpenCV / ORB-SLAM steering), all 28 bytes of every keypoint and all 256 bits of every descriptor against the oracle's
    steps 6' / 8' -- with the built-in pattern and with one that has the reach of the canonical ORB table (radius 18.4);
    keypoints 19..24 px from the border read mirrored pixels."""
    import torch
    from gslam_amd.orb import OrbExtractor, kps_to_numpy, synth_frames
    ex = OrbExtractor(ctx, w, h, max_batch=B, n_features=K)
    ex.set_steering(1)
    pat = _steer_pattern() if pattern else None
    if pat is not None:
        ex.set_pattern(pat)
    frames = synth_frames(ctx, B, w, h, base_seed=0x5EED0300 + w)
    kps, desc, counts = ex.extract(frames)
    torch.cuda.synchronize()
    host = frames.cpu().numpy()[:, :, :w]
    oracle.orb_set_steer(1)
    try:
        assert pat is None or oracle.orb_set_pattern(pat)
        near = 0
        for f in range(B):
            ek, ed = oracle.orb_extract(host[f], K)
            n = int(counts[f])
            assert n == len(ek)
            assert kps_to_numpy(kps)[f, :n].tobytes() == ek.tobytes(), f"frame {f}: keypoint records differ"
            assert np.array_equal(desc[f, :n].cpu().numpy(), ed), f"frame {f}: descriptor bits differ"
            lv0 = ek[ek["octave"] == 0]
            near += int(((lv0["x"] < 22) | (lv0["y"] < 22) | (lv0["x"] > w - 23) | (lv0["y"] > h - 23)).sum())
            assert len(np.unique(ek["angle"])) > 50  # continuous, not 30 bins
        assert near > 0, "no keypoint exercised the mirrored border"
    finally:
        oracle.orb_set_pattern(None)
        oracle.orb_set_steer(0)
    ex.close()


def test_steering_mode_rules_and_vocabulary_round_trip(ctx, oracle):
    """(a) the 30-bin mode refuses a pattern of radius 18 and the continuous mode takes it; going back to bins with it
    installed is refused; switching modes changes descriptors and angles, not keypoint positions; (b) the steered
    descriptors go through the GPU vocabulary (gh_bow_transform_dev) to the same BoW vector as the oracle's transform."""
    import torch
    from gslam_amd import bow_synth, hip
    from gslam_amd.bow import Vocabulary
    from gslam_amd.orb import OrbExtractor, kps_to_numpy, synth_frames
    pat = _steer_pattern()
    ex = OrbExtractor(ctx, 640, 480, max_batch=1, n_features=800)
    frames = synth_frames(ctx, 1, 640, 480, base_seed=0x5EED0400)
    out = ex.alloc_outputs(1)
    k0, d0, c0 = [t.clone() for t in ex.extract(frames, out)]
    with pytest.raises(hip.GslamHipError):
        ex.set_pattern(pat)
    ex.set_steering(1)
    ex.set_pattern(pat)
    with pytest.raises(hip.GslamHipError):
        ex.set_steering(0)
    k1, d1, c1 = ex.extract(frames, out)  # the same buffers: a graph captured for the other mode must not be replayed
    torch.cuda.synchronize()
    a0, a1 = kps_to_numpy(k0)[0], kps_to_numpy(k1)[0]
    assert torch.equal(c0, c1)
    for f in ("x", "y", "size", "response", "octave"):
        assert np.array_equal(a0[f], a1[f])
    assert not np.array_equal(a0["angle"], a1["angle"]) and not torch.equal(d0, d1)
    d = (a1["angle"] - a0["angle"] + 180.0) % 360.0 - 180.0
    assert np.abs(d).max() <= 12.5  # the bin is the 12-degree sector that holds the continuous angle
    voc = bow_synth.make_vocabulary(k=10, L=4, seed=2)
    v = Vocabulary(ctx, voc)
    n = int(c1[0])
    word, weight, node, bw, bv, bn = v.transform(d1, c1, 3)
    torch.cuda.synchronize()
    ew, ewt, en, ebw, ebv = oracle.bow_transform(voc, d1[0, :n].cpu().numpy(), 3)
    assert np.array_equal(word[0, :n].cpu().numpy().astype(np.uint32), ew) and np.array_equal(node[0, :n].cpu().numpy().astype(np.uint32), en)
    m = int(bn[0])
    assert m == len(ebw) and np.array_equal(bw[0, :m].cpu().numpy().astype(np.uint32), ebw) and np.array_equal(bv[0, :m].cpu().numpy(), ebv)
    ex.close()


# ---------------------------------------------------------------- quadtree distribution (gh_orb_plan_set_distribution)
def _quadtree_case(ctx, oracle, host, K, steer, **kw):
    import torch
    from gslam_amd.orb import OrbExtractor, kps_to_numpy
    B, h, w = host.shape
    ex = OrbExtractor(ctx, w, h, max_batch=B, n_features=K, **kw)
    ex.set_distribution(1)
    if steer:
        ex.set_steering(1)
    frames = torch.from_numpy(np.ascontiguousarray(host)).cuda()
    kps, desc, counts = ex.extract(frames)
    torch.cuda.synchronize()
    kps, desc, counts = kps_to_numpy(kps), desc.cpu().numpy(), counts.cpu().numpy()
    okw = {}
    if "n_levels" in kw:
        okw["nlevels"] = kw["n_levels"]
    if "ini_th" in kw:
        okw["ini_th"], okw["min_th"] = kw["ini_th"], kw["min_th"]
    oracle.orb_set_distribution(1)
    oracle.orb_set_steer(1 if steer else 0)
    try:
        for f in range(B):
            ek, ed = oracle.orb_extract(host[f], K, **okw)
            n = len(ek)
            assert counts[f] == n, f"frame {f}: count {counts[f]} vs oracle {n}"
            assert kps[f, :n].tobytes() == ek.tobytes(), f"frame {f}: keypoint records differ"
            assert np.array_equal(desc[f, :n], ed), f"frame {f}: descriptor bits differ"
            assert not kps[f, n:].tobytes().strip(b"\0") and not desc[f, n:].any()
    finally:
        oracle.orb_set_distribution(0)
        oracle.orb_set_steer(0)
    # back to the default mode on the same plan: the default results
    ex.set_distribution(0)
    ex.set_steering(0)
    k0, d0, c0 = ex.extract(frames)
    torch.cuda.synchronize()
    ek, ed = oracle.orb_extract(host[0], K, **okw)
    assert int(c0[0]) == len(ek) and kps_to_numpy(k0)[0, :len(ek)].tobytes() == ek.tobytes()
    ex.close()
    return counts


@pytest.mark.parametrize("w,h,K,B,steer", [(640, 480, 1000, 3, False), (640, 480, 2000, 2, True), (752, 480, 1500, 2, True),
                                           (333, 257, 300, 3, False), (161, 123, 200, 2, True), (1241, 376, 2000, 2, True),
                                           (150, 420, 400, 2, False), (1920, 1080, 2000, 2, True)])
def test_quadtree_distribution_parity(ctx, oracle, w, h, K, B, steer):
    """gh_orb_plan_set_distribution(plan, 1): ORB-SLAM's per-cell FAST + DistributeOctTree (oracle steps 4', 5'), all 28
    bytes of every keypoint and every descriptor bit against the oracle -- VGA, KITTI's aspect ratio (4 roots), a portrait
    frame (one root), 1080p, with and without continuous steering."""
    from gslam_amd.orb import synth_frames
    host = synth_frames(ctx, B, w, h, base_seed=0x5EED0400 + w).cpu().numpy()[:, :, :w]
    counts = _quadtree_case(ctx, oracle, host, K, steer)
    assert (counts > 0).all()


@pytest.mark.parametrize("nodes", ["512", "1024", "2048"])
def test_quadtree_tree_kernel_table_capacities(ctx, oracle, monkeypatch, nodes):
    """Round 5b: the tree kernel is compiled for node tables of 512 / 1024 / 2048 entries (256 / 512 / 1024 threads) and picks
    by quota and launch size; here every capacity is forced (GSLAM_HIP_QT_NODES, read when the plan switches mode) on frames
    whose levels take both cell kernels' size classes (VGA: cells of 31 .. 35 pixels), clean and noisy."""
    from gslam_amd.orb import synth_frames
    monkeypatch.setenv("GSLAM_HIP_QT_NODES", nodes)
    host = synth_frames(ctx, 2, 640, 480, base_seed=0x5EED0777).cpu().numpy()[:, :, :640]
    _quadtree_case(ctx, oracle, host, 1000, True)
    rng = np.random.default_rng(int(nodes))
    noise = rng.integers(0, 256, (1, 260, 380), dtype=np.uint8)
    _quadtree_case(ctx, oracle, noise, 300, False)


def test_quadtree_distribution_dense_candidates_and_small_quotas(ctx, oracle):
    """Noise frames: ~1 candidate per 12 pixels, deep trees, stage (B) of the tree repeated; K from 8 (quotas of 1-2 per
    level, the first pass already overshoots them) to 6000; other level counts and thresholds."""
    rng = np.random.default_rng(77)
    host = rng.integers(0, 256, (2, 300, 420), dtype=np.uint8)
    for K in (8, 60, 1000, 6000):
        _quadtree_case(ctx, oracle, host, K, False)
    _quadtree_case(ctx, oracle, host, 700, True, n_levels=3)
    _quadtree_case(ctx, oracle, host, 700, False, n_levels=5, ini_th=40, min_th=15)
    flat = np.full((1, 200, 320), 90, np.uint8)
    flat[0, 100, 160] = 255  # a single corner-like blob
    c = _quadtree_case(ctx, oracle, flat, 100, False)
    assert c[0] <= 8


def test_quadtree_distribution_limits_and_host_entry(ctx, oracle):
    """quota > 2045 per level is refused loudly; the single-frame host entry point honours the mode."""
    from gslam_amd import hip
    from gslam_amd.orb import OrbExtractor
    ex = OrbExtractor(ctx, 640, 480, max_batch=1, n_features=12000)
    with pytest.raises(hip.GslamHipError):
        ex.set_distribution(1)
    ex.close()
    ex = OrbExtractor(ctx, 640, 480, max_batch=1, n_features=800)
    ex.set_distribution(1)
    img = oracle.synth_frame(640, 480, 1234)
    kps, desc = ex.extract_host(img)
    oracle.orb_set_distribution(1)
    try:
        ek, ed = oracle.orb_extract(img, 800)
    finally:
        oracle.orb_set_distribution(0)
    assert kps.tobytes() == ek.tobytes() and np.array_equal(desc, ed)
    ex.close()

def test_quadtree_plane_by_sliding_window_kernel(ctx, oracle, monkeypatch):
    """Round-6 experiment (profiles/orb_sliding_window_r06.txt): the score plane by the barrier-free sliding-window kernel
    (GSLAM_HIP_ORB_PLANE_SW, read per call; variant bits 2 / 4 / 8 = the A/B forms) -- same keypoints and descriptors as the oracle,
    i.e. as the tile kernel's plane, on images whose width / height leave partial strips and partial row groups."""
    for var, (w, h, B) in (("1", (640, 480, 3)), ("15", (752, 480, 2)), ("1", (1241, 376, 2)), ("7", (333, 259, 2))):
        from gslam_amd.orb import synth_frames
        monkeypatch.setenv("GSLAM_HIP_ORB_PLANE_SW", var)
        host = synth_frames(ctx, B, w, h, base_seed=0x5EED0900 + w).cpu().numpy()[:, :, :w]
        counts = _quadtree_case(ctx, oracle, host, 1000, False)
        assert (counts > 0).all()
    # candidate density at its worst (every queue chunk full, the group's leftover carried): noise
    monkeypatch.setenv("GSLAM_HIP_ORB_PLANE_SW", "1")
    rng = np.random.default_rng(5)
    _quadtree_case(ctx, oracle, rng.integers(0, 256, (2, 300, 520), dtype=np.uint8), 500, False)
