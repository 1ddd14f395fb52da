"""GPU parity of the undistortion remap (HIP, through the C ABI) vs the oracle pinned to the reference's own
Undistorter, and vs the reference's golden outputs directly: bit-exact wherever the reference defines the pixel."""
import os

import numpy as np
import pytest

from gslam_amd import undist_synth

pytestmark = pytest.mark.gpu


def _img(h, w, ch, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (h, w) if ch == 1 else (h, w, ch), dtype=np.uint8)


def test_undistort_golden_reference_vectors(ctx, oracle):
    from gslam_amd.undist import Undistorter
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "undist_reference.npz"))
    t = {k: g[k] for k in ("remapX", "remapFast", "remapIdx", "remapCoef")}
    t.update(w_in=int(g["w_in"]), h_in=int(g["h_in"]), w_out=int(g["w_out"]), h_out=int(g["h_out"]))
    u = Undistorter(ctx, t)
    for ch in (1, 3):
        for fast in (0, 1):
            got = u.undistort_host(g[f"img{ch}"], fast=bool(fast))
            eo, wr = oracle.undistort(g[f"img{ch}"], t, fast=bool(fast))
            assert np.array_equal(got, eo)                       # every byte equals the oracle (incl. zero fill)
            assert np.array_equal(got[wr], g[f"out{ch}_{fast}"][wr])  # and the reference where it is defined
    u.close()


@pytest.mark.parametrize("ch,fast", [(1, False), (1, True), (3, False), (3, True), (4, True)])
def test_undistort_batched_parity(ctx, oracle, ch, fast):
    import torch
    from gslam_amd.undist import Undistorter
    t = undist_synth.make_tables(640, 480, 600, 440)
    u = Undistorter(ctx, t)
    imgs = np.stack([_img(480, 640, ch, 10 + b) for b in range(3)])
    out = u.undistort(torch.from_numpy(imgs).cuda(), fast=fast)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    for b in range(3):
        eo, _ = oracle.undistort(imgs[b], t, fast=fast)
        assert np.array_equal(out[b], eo)
    u.close()
