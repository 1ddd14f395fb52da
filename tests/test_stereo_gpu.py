"""BASELINE configs[2] ("C3", KITTI-like stereo 1241x376 x 2): per-frame extraction of both eyes + row-band
restricted left-right matching + temporal matching, GPU vs oracle, bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_stereo_extract_band_match_and_temporal_match(ctx, oracle):
    import torch
    from gslam_amd.matcher import BFMatcher
    from gslam_amd.orb import OrbExtractor, kps_to_numpy
    W, H, K, T = 1241, 376, 2000, 2
    # right eye = left eye shifted by a disparity of 24 px (plus its own noise columns at the border)
    frames = []
    for t in range(T):
        wide = oracle.synth_frame(W + 24, H, 0xC3000 + t)
        frames += [wide[:, 24:], wide[:, :W]]  # left, right
    frames = np.stack(frames)  # L0 R0 L1 R1
    ex = OrbExtractor(ctx, W, H, max_batch=len(frames), n_features=K)
    buf = np.zeros((len(frames), H, 1244), np.uint8)  # dword-aligned stride
    buf[:, :, :W] = frames
    kps, desc, counts = ex.extract(torch.from_numpy(buf).cuda())
    torch.cuda.synchronize()
    kp_np, desc_np, cnt = kps_to_numpy(kps), desc.cpu().numpy(), counts.cpu().numpy()
    for f in range(len(frames)):
        ek, ed = oracle.orb_extract(frames[f], K)
        assert cnt[f] == len(ek) and kp_np[f, :cnt[f]].tobytes() == ek.tobytes() and np.array_equal(desc_np[f, :cnt[f]], ed)
    m = BFMatcher(ctx)
    band = 2.0 / 31.0  # +-2 px at level 0, growing with the level scale
    pq = torch.tensor([0, 2], dtype=torch.int32).cuda()  # L_t -> R_t
    pt = torch.tensor([1, 3], dtype=torch.int32).cuda()
    idx1, d1, d2 = m.match_band_pairs(desc, kps, counts, pq, pt, band)
    tq = torch.tensor([0], dtype=torch.int32).cuda()      # temporal L_0 -> L_1 (plain BF)
    tt = torch.tensor([2], dtype=torch.int32).cuda()
    t_idx, t_d1, t_d2 = m.match_pairs(desc, counts, tq, tt)
    torch.cuda.synchronize()
    idx1, d1, d2 = idx1.cpu().numpy(), d1.cpu().numpy().view(np.uint16), d2.cpu().numpy().view(np.uint16)
    for p, (a, b) in enumerate([(0, 1), (2, 3)]):
        na, nb = cnt[a], cnt[b]
        e = oracle.bf_match_band(desc_np[a, :na], kp_np[a, :na], desc_np[b, :nb], kp_np[b, :nb], band)
        assert np.array_equal(idx1[p, :na], e[0]) and np.array_equal(d1[p, :na], e[1]) and np.array_equal(d2[p, :na], e[2])
        # the synthetic 24 px disparity is recovered by the exact-descriptor matches
        ok = (e[1] == 0) & (kp_np[a, :na]["octave"] == 0)
        dx = kp_np[a, :na]["x"][ok] - kp_np[b, :nb]["x"][e[0][ok]]
        assert ok.sum() > 100 and (dx == -24).mean() > 0.95
        assert (e[0] == -1).sum() < na  # band leaves candidates
    e = oracle.bf_match(desc_np[0, :cnt[0]], desc_np[2, :cnt[2]], threads=4)
    assert np.array_equal(t_idx.cpu().numpy()[0, :cnt[0]], e[0])
    ex.close()
