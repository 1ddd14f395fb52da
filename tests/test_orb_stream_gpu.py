"""gh_orb_stream_*: the host-fed extraction path of the C ABI (frames in host memory -> packed records in pinned host
memory, three HIP streams, no torch).  Every record of every frame must equal the oracle bit for bit -- across ring
wrap-around, partial chunks, the staging block and caller-owned buffers, padded strides, BGR input, and a producer thread
submitting while a consumer thread collects.  Serves GSLAM/plugins/play/main.cpp:99-155 (frames published one by one)."""
import threading

import numpy as np
import pytest

from orb_images import CLASSES

pytestmark = pytest.mark.gpu


def _expect(oracle, frames, K, **kw):
    ks, ds = [], []
    for f in frames:
        k, d = oracle.orb_extract(f, K, **kw)
        ks.append(k)
        ds.append(d)
    return ks, ds


def _check_chunk(res, ks, ds):
    off, kps, desc, gpu_ms = res
    assert len(off) == len(ks) + 1 and off[0] == 0
    for f in range(len(ks)):
        n = len(ks[f])
        assert off[f + 1] - off[f] == n, (f, off[f + 1] - off[f], n)
        assert kps[off[f]:off[f + 1]].tobytes() == ks[f].tobytes(), f"frame {f}: keypoints differ"
        assert np.array_equal(desc[off[f]:off[f + 1]], ds[f]), f"frame {f}: descriptors differ"
    assert gpu_ms > 0


def test_stream_ring_wraparound_and_partial_chunks(ctx, oracle):
    from gslam_amd.orb import OrbStream
    w, h, K, chunk, depth = 640, 480, 800, 4, 3
    names = ["noise", "few_corners", "mixed", "step_edges", "black", "dots8"]
    frames = [oracle.synth_frame(w, h, 900 + i) if i % 2 == 0 else CLASSES[names[(i // 2) % len(names)]](w, h, i)
              for i in range(27)]
    ks, ds = _expect(oracle, frames, K)
    st = OrbStream(ctx, w, h, chunk, depth, n_features=K)
    tickets, pos = [], 0
    sizes = [4, 4, 1, 4, 3, 4, 2, 4, 1]  # 27 frames over 9 tickets: the ring of 3 wraps three times
    results = {}
    for i, n in enumerate(sizes):
        if i % 2 == 0:  # through the pinned staging block
            buf = st.staging()
            for j in range(n):
                buf[j, : w * h] = frames[pos + j].reshape(-1)
            t = st.submit(None, n)
        else:           # from the caller's own (pageable) memory
            t = st.submit(np.ascontiguousarray(np.stack(frames[pos:pos + n])))
        tickets.append((t, pos, n))
        pos += n
        if len(tickets) - len(results) == depth:  # collect the oldest before its slot is reused
            t0, p0, n0 = tickets[len(results)]
            results[t0] = st.collect(t0)
            _check_chunk(results[t0], ks[p0:p0 + n0], ds[p0:p0 + n0])
    for t0, p0, n0 in tickets[len(results):]:
        _check_chunk(st.collect(t0), ks[p0:p0 + n0], ds[p0:p0 + n0])
    # a ticket that left the ring is refused, not silently wrong
    with pytest.raises(Exception):
        st.collect(tickets[0][0])
    st.close()


def test_stream_padded_strides_and_bgr(ctx, oracle):
    from gslam_amd.orb import OrbStream
    w, h, K = 333, 257, 500
    rng = np.random.default_rng(5)
    # gray, row stride 340 (not a multiple of 16 -> the extractor stages level 0), frames 7 bytes further apart
    rs, fs = 340, 340 * h + 7
    st = OrbStream(ctx, w, h, 3, 2, row_stride=rs, frame_stride=fs, n_features=K)
    frames = [oracle.synth_frame(w, h, 40 + i) for i in range(3)]
    buf = st.staging()
    buf[:] = 0xAB  # padding bytes must not matter
    for j, f in enumerate(frames):
        v = buf[j, : rs * h].reshape(h, rs)
        v[:, :w] = f
    _check_chunk(st.collect(st.submit(None, 3)), *_expect(oracle, frames, K))
    st.close()
    # BGR and BGRA: fixed-point luma on the device (same rule as gh_bgr_to_gray_dev / the oracle)
    for ch in (3, 4):
        st = OrbStream(ctx, w, h, 2, 2, channels=ch, n_features=K)
        imgs = [rng.integers(0, 256, (h, w, ch), dtype=np.uint8) for _ in range(2)]
        for im in imgs:  # structure, so that there are corners
            im[..., :3] = (im[..., :3] // 4 + oracle.synth_frame(w, h, 77)[..., None] // 2).astype(np.uint8)
        gray = [oracle.bgr_to_gray(im) for im in imgs]
        t = st.submit(np.ascontiguousarray(np.stack(imgs)).reshape(2, -1))
        _check_chunk(st.collect(t), *_expect(oracle, gray, K))
        st.close()


def test_stream_producer_consumer_threads(ctx, oracle):
    """A producer thread submits while a consumer thread collects (collect does not hold the context lock)."""
    from gslam_amd.orb import OrbStream
    w, h, K, chunk, depth, n_chunks = 640, 480, 1000, 2, 3, 12
    frames = [oracle.synth_frame(w, h, 5000 + i) for i in range(chunk * n_chunks)]
    ks, ds = _expect(oracle, frames, K)
    st = OrbStream(ctx, w, h, chunk, depth, n_features=K)
    sem = threading.Semaphore(depth)  # never more than `depth` uncollected tickets
    q, errs = [], []
    cv = threading.Condition()

    def producer():
        try:
            for c in range(n_chunks):
                sem.acquire()
                t = st.submit(np.ascontiguousarray(np.stack(frames[c * chunk:(c + 1) * chunk])))
                with cv:
                    q.append((t, c))
                    cv.notify()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            with cv:
                q.append(None)
                cv.notify()

    def consumer():
        try:
            for _ in range(n_chunks):
                with cv:
                    while not q:
                        cv.wait()
                    item = q.pop(0)
                if item is None:
                    return
                t, c = item
                _check_chunk(st.collect(t), ks[c * chunk:(c + 1) * chunk], ds[c * chunk:(c + 1) * chunk])
                sem.release()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            for _ in range(depth):
                sem.release()

    tp, tc = threading.Thread(target=producer), threading.Thread(target=consumer)
    tp.start(); tc.start(); tp.join(120); tc.join(120)
    assert not errs, errs
    assert not tp.is_alive() and not tc.is_alive()
    st.close()


def test_stream_matches_resident_path_at_1080p(ctx, oracle):
    """Bench geometry: the stream's packed output equals gh_orb_extract_dev's on the same frames."""
    import torch
    from gslam_amd.orb import OrbExtractor, OrbStream, kps_to_numpy, synth_frames
    w, h, K, n = 1920, 1080, 2000, 6
    dev = synth_frames(ctx, n, w, h, base_seed=0x5EED0000)
    ex = OrbExtractor(ctx, w, h, max_batch=n, n_features=K)
    kps, desc, counts = ex.extract(dev)
    torch.cuda.synchronize()
    kp_np, d_np, c_np = kps_to_numpy(kps), desc.cpu().numpy(), counts.cpu().numpy()
    host = dev.cpu().numpy().reshape(n, -1)
    st = OrbStream(ctx, w, h, 3, 2, n_features=K)
    t0, t1 = st.submit(host[:3]), st.submit(host[3:])
    for t, base in ((t0, 0), (t1, 3)):
        off, k, d, _ = st.collect(t)
        for f in range(3):
            c = c_np[base + f]
            assert off[f + 1] - off[f] == c
            assert k[off[f]:off[f + 1]].tobytes() == kp_np[base + f, :c].tobytes()
            assert np.array_equal(d[off[f]:off[f + 1]], d_np[base + f, :c])
    st.close()
    ex.close()


def test_featuredetector_batch_and_async_entries(tmp_path, oracle):
    """FeatureDetector::detectAndComputeBatch and the asynchronous submit / collect pair (gslam_amd/plugin/FeatureDetector.h)
    inside a real GSLAM host process: per-frame, batch and async results of the same 13 frames all equal the oracle."""
    import os
    import struct
    import subprocess
    import oracle_lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host, libdir = os.path.join(root, "build", "plugin_host"), os.path.join(root, "gslam_amd", "lib")
    if not (os.path.exists(host) and os.path.exists(os.path.join(libdir, "libgslam_featuredetector.so"))):
        pytest.skip("build/plugin_host or libgslam_featuredetector.so missing (run `make plugins` where the GSLAM headers are)")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = libdir + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    for ch in (1, 3):
        w, h, K, n = 320, 240, 600, 13
        gray = [oracle.synth_frame(w, h, 300 + i) if i % 3 else CLASSES["noise"](w, h, i) for i in range(n)]
        if ch == 1:
            imgs, expect_gray = np.stack(gray), gray
        else:
            rng = np.random.default_rng(1)
            imgs = np.stack([np.stack([g // 2 + rng.integers(0, 100, g.shape, dtype=np.uint8) for _ in range(3)], axis=-1)
                             for g in gray]).astype(np.uint8)
            expect_gray = [oracle.bgr_to_gray(im) for im in imgs]
        inp, out = tmp_path / f"frames{ch}.raw", tmp_path / f"out{ch}.bin"
        imgs.tofile(inp)
        r = subprocess.run([host, "orbbatch", libdir, str(w), str(h), str(ch), str(n), str(inp), str(out), str(K)],
                           capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "matchBatch=1" in r.stdout  # FeatureDetector::matchBatch (all ordered frame pairs, one device batch) == match() per pair
        raw = open(out, "rb").read()
        ok, okb, oka, nn = struct.unpack("4i", raw[:16])
        assert (ok, okb, oka, nn) == (1, 1, 1, n)
        pos = 16
        for which in ("single", "batch", "async"):
            for f in range(n):
                ek, ed = oracle.orb_extract(expect_gray[f], K)
                (m,) = struct.unpack("i", raw[pos:pos + 4])
                pos += 4
                assert m == len(ek), (which, f, m, len(ek))
                assert raw[pos:pos + 28 * m] == ek.tobytes(), (which, f)
                pos += 28 * m
                assert raw[pos:pos + 32 * m] == ed.tobytes(), (which, f)
                pos += 32 * m
        assert pos == len(raw)
