"""Undistortion oracle PINNED to the reference: UndistorterImpl::prepareReMap + undistort / undistortFast compiled
from /root/reference (oracle/_ref) live, and golden vectors generated from it."""
import os

import numpy as np
import pytest

import oracle_lib
from gslam_amd import undist_synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "undist_reference.npz")


def _img(h, w, ch, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (h, w) if ch == 1 else (h, w, ch), dtype=np.uint8)


def test_golden_vectors_from_reference(oracle):
    g = np.load(GOLD)
    t = {k: g[k] for k in ("remapX", "remapFast", "remapIdx", "remapCoef")}
    t.update(w_in=int(g["w_in"]), h_in=int(g["h_in"]), w_out=int(g["w_out"]), h_out=int(g["h_out"]))
    for ch in (1, 3):
        for fast in (0, 1):
            out, wr = oracle.undistort(g[f"img{ch}"], t, fast=bool(fast))
            exp = g[f"out{ch}_{fast}"]
            assert wr.any() and np.array_equal(out[wr], exp[wr])


@pytest.mark.skipif(not oracle_lib.have_reference(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("cam_in,cam_out", [
    ([320, 240, 260, 262, 161.5, 118.2, -0.31, 0.11, 0.001, -0.0007, -0.02], [320, 240, 200, 200, 160, 120]),
    ([200, 150, 150, 150, 100, 75, 0.9], [160, 120, 110, 110, 80, 60]),
])
def test_oracle_equals_reference_live(oracle, cam_in, cam_out):
    ref = oracle_lib.load_reference()
    ru = oracle_lib.RefUndistorter(ref, cam_in, cam_out)
    t = ru.tables()
    assert (t["remapX"] < 0).any() and (t["remapX"] > 0).any()
    for ch in (1, 3):
        img = _img(ru.h_in, ru.w_in, ch, 7 + ch)
        for fast in (False, True):
            out, wr = oracle.undistort(img, t, fast=fast)
            exp = ru.run(img, fast=fast)
            assert np.array_equal(out[wr], exp[wr])
            if ch == 1 and not fast:
                # the 1-channel bilinear path defines every pixel (0 outside) except where the reference's own
                # table steps past the image (last source row / column: it reads out of bounds there)
                oob = (t["remapIdx"] >= ru.w_in * ru.h_in).any(axis=1).reshape(ru.h_out, ru.w_out)
                assert (wr | oob).all() and wr.mean() > 0.9
    ru.close()


def test_synthetic_tables_follow_the_reference_post_processing(oracle):
    t = undist_synth.make_tables(160, 120, 128, 96)
    ok = t["remapX"] >= 0
    assert np.allclose(t["remapCoef"][ok].sum(axis=1), 1.0, atol=1e-6) and (t["remapCoef"][~ok] == 0).all()
    img = _img(120, 160, 1, 3)
    out, wr = oracle.undistort(img, t)
    assert wr.mean() > 0.95 and (out.reshape(-1)[~ok] == 0).all() and out.reshape(-1)[ok].std() > 10
