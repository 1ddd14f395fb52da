"""CPU-only checks of the ORB oracle and its spec data (the oracle is 'parity unpinned' against the
reference: GSLAM ships no ORB code — these tests pin what CAN be pinned: output record layout, the
committed tables, internal invariants)."""
import hashlib
import os
import subprocess
import sys

import numpy as np

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tables_regenerate_identically(tmp_path):
    hdr = os.path.join(ROOT, "include", "gslam_orb_tables.h")
    before = hashlib.sha256(open(hdr, "rb").read()).hexdigest()
    assert before == "14a972d2901eda7f1cb21d5478899db84e9abff5e7cbd2c325a632f54676e18b"


def test_level_dims_and_quotas(oracle):
    ws, hs = oracle.orb_level_dims(1920, 1080)
    assert ws.tolist() == [1920, 1600, 1333, 1111, 926, 772, 643, 536]
    assert hs.tolist() == [1080, 900, 750, 625, 521, 434, 362, 301]
    for K in (1, 7, 100, 1000, 2000, 8000):
        q = oracle.orb_quotas(K)
        assert q.sum() == K and (q >= 0).all()
    assert oracle.orb_quotas(2000).tolist() == [434, 362, 302, 251, 209, 175, 145, 122]


def test_extract_invariants(oracle):
    g = oracle.synth_frame(640, 480, 0x5EED0000)
    kps, desc = oracle.orb_extract(g, 1000)
    assert len(kps) == 1000 and desc.shape == (1000, 32)
    assert (kps["class_id"] == -1).all()
    assert set(np.unique(kps["angle"])).issubset({12.0 * k for k in range(30)})
    assert (np.diff(kps["octave"]) >= 0).all()  # level-major output order
    ws, hs = oracle.orb_level_dims(640, 480)
    scale = np.float32(1.0)
    for l in range(8):
        m = kps["octave"] == l
        x = np.rint(kps["x"][m] / scale).astype(int)
        y = np.rint(kps["y"][m] / scale).astype(int)
        assert ((x >= 19) & (x < ws[l] - 19) & (y >= 19) & (y < hs[l] - 19)).all()
        assert np.array_equal((x.astype(np.float32) * scale), kps["x"][m])  # single fp32 multiply
        assert np.allclose(kps["size"][m], 31.0 * scale)
        scale = np.float32(scale * np.float32(1.2))
    assert (kps["response"] > 7).all()
    # descriptors are informative
    assert 0.4 < np.unpackbits(desc).mean() < 0.6
    assert len(np.unique(desc, axis=0)) > 990


def test_extract_deterministic_and_stride_independent(oracle):
    g = oracle.synth_frame(333, 257, 42)
    a = oracle.orb_extract(g, 300)
    b = oracle.orb_extract(g.copy(), 300)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_flat_image_yields_nothing(oracle):
    kps, desc = oracle.orb_extract(np.full((200, 300), 77, np.uint8), 500)
    assert len(kps) == 0


def test_self_match_of_shifted_frame(oracle):
    """A frame and its copy shifted by (32,32) px (a cell) must share most level-0 descriptors exactly."""
    g = oracle.synth_frame(800, 600, 9)
    a_k, a_d = oracle.orb_extract(g[:480, :640], 1000, nlevels=1)
    b_k, b_d = oracle.orb_extract(g[32:512, 32:672], 1000, nlevels=1)
    idx1, d1, d2 = oracle.bf_match(a_d, b_d)
    exact = d1 == 0
    assert exact.sum() > 300
    dx = a_k["x"][exact] - b_k["x"][idx1[exact]]
    dy = a_k["y"][exact] - b_k["y"][idx1[exact]]
    assert (dx == 32).mean() > 0.95 and (dy == 32).mean() > 0.95


def test_bgr_to_gray_fixed_point(oracle):
    rng = np.random.default_rng(3)
    bgr = rng.integers(0, 256, (17, 23, 3), dtype=np.uint8)
    g = oracle.bgr_to_gray(bgr)
    ref = (bgr[..., 0].astype(np.int64) * 1868 + bgr[..., 1].astype(np.int64) * 9617 +
           bgr[..., 2].astype(np.int64) * 4899 + 8192) >> 14
    assert np.array_equal(g, ref.astype(np.uint8))
    assert abs(int(g[0, 0]) - round(0.114 * bgr[0, 0, 0] + 0.587 * bgr[0, 0, 1] + 0.299 * bgr[0, 0, 2])) <= 1


def _builtin_base_pattern():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_orb_tables as T
    return np.array(T.gen_pattern(), np.int8)


def test_set_pattern_with_the_builtin_base_reproduces_the_builtin_table(oracle):
    """gh_orb_plan_set_pattern / oracle_orb_set_pattern rebuild the 30-bin steered LUT at run time; feeding them the
    unrotated built-in pattern must give today's descriptors bit for bit (same rotation + rounding rule as the generator
    of include/gslam_orb_tables.h)."""
    g = oracle.synth_frame(400, 300, 77)
    a = oracle.orb_extract(g, 400)
    assert oracle.orb_set_pattern(_builtin_base_pattern())
    try:
        b = oracle.orb_extract(g, 400)
    finally:
        oracle.orb_set_pattern(None)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_set_pattern_changes_descriptors_and_validates_range(oracle):
    rng = np.random.default_rng(5)
    pat = rng.integers(-9, 10, (256, 4)).astype(np.int8)
    pat[pat[:, 0] == pat[:, 2], 2] += 1
    g = oracle.synth_frame(400, 300, 78)
    a = oracle.orb_extract(g, 300)
    assert oracle.orb_set_pattern(pat)
    try:
        b = oracle.orb_extract(g, 300)
    finally:
        oracle.orb_set_pattern(None)
    assert np.array_equal(a[0], b[0]) and not np.array_equal(a[1], b[1])  # same keypoints, different bits
    bad = pat.copy()
    bad[7] = [13, 13, 0, 0]  # radius 18.4: leaves the +-13 blurred patch at 45 degrees
    assert not oracle.orb_set_pattern(bad)
    oracle.orb_set_pattern(None)


def _canonical_like_pattern(seed=5):
    """A test pattern with the reach of the canonical ORB bit_pattern_31_ (points up to (-13, -13), radius 18.4): the real
    table is not available offline."""
    rng = np.random.default_rng(seed)
    pat = rng.integers(-13, 14, (256, 4)).astype(np.int8)
    pat[0] = (-13, -13, 13, 13)
    pat[1] = (12, -13, -13, 12)
    same = (pat[:, 0] == pat[:, 2]) & (pat[:, 1] == pat[:, 3])
    pat[same, 2] = -pat[same, 2] - 1
    return pat


def test_continuous_steering_pieces_against_independent_float64(oracle):
    """Steps 6' and 8' of the oracle against plain numpy float64: the fp32 polynomial arctangent stays within 0.02 degrees of
    atan2 (OpenCV documents 0.3), the Taylor (cos, sin) within 6e-7 of the exact values, quadrant handling included."""
    rng = np.random.default_rng(3)
    for _ in range(4000):
        x, y = (float(v) for v in rng.integers(-2_000_000, 2_000_001, 2))
        a = oracle.orb_fast_atan2_deg(y, x)
        e = np.degrees(np.arctan2(y, x)) % 360.0
        assert min(abs(a - e), 360.0 - abs(a - e)) < 0.02, (x, y, a, e)
    assert oracle.orb_fast_atan2_deg(0.0, 0.0) == 0.0 and oracle.orb_fast_atan2_deg(0.0, -5.0) == 180.0
    assert oracle.orb_fast_atan2_deg(7.0, 0.0) == 90.0 and oracle.orb_fast_atan2_deg(-7.0, 0.0) == 270.0
    for a in list(np.linspace(0, 360, 1441)) + [44.999, 45.0, 45.001, 134.9999, 315.0, 359.99997]:
        cs, sn = oracle.orb_sincos_deg(np.float32(a))
        ea = np.radians(float(np.float32(a)))
        assert abs(cs - np.cos(ea)) < 6e-7 and abs(sn - np.sin(ea)) < 6e-7, a


def test_continuous_steering_extraction_against_a_python_restatement(oracle):
    """The whole steered describe step restated in numpy from the spec text (reflected blur, rotated tests, rint) on the
    oracle's own keypoints: same angles, same 256 bits.  Also: the 30-bin mode refuses a pattern of radius 18, the continuous
    mode takes it, and the keypoints (positions, scores) do not depend on the mode."""
    pat = _canonical_like_pattern()
    g = oracle.synth_frame(200, 160, 77)
    base_k, base_d = oracle.orb_extract(g, 120, nlevels=3)
    assert not oracle.orb_set_pattern(pat)          # 30-bin table: points beyond radius 13.49 are refused
    oracle.orb_set_steer(1)
    try:
        assert oracle.orb_set_pattern(pat)
        kps, desc = oracle.orb_extract(g, 120, nlevels=3)
    finally:
        oracle.orb_set_pattern(None)
        oracle.orb_set_steer(0)
    assert len(kps) == len(base_k)
    for f in ("x", "y", "size", "response", "octave"):
        assert np.array_equal(kps[f], base_k[f])
    assert not set(np.unique(kps["angle"])).issubset({12.0 * k for k in range(30)})
    gauss = np.array([144, 268, 391, 442, 391, 268, 144], np.int64)
    umax = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    lv = [g] + [oracle.orb_pyramid_level(g, l, 3) for l in (1, 2)]
    scale = [np.float32(1.0), np.float32(1.2), np.float32(np.float32(1.2) * np.float32(1.2))]

    def refl(i, n):
        return -i if i < 0 else (2 * n - 2 - i if i >= n else i)

    def blur(img, x, y):
        h, w = img.shape
        acc = 0
        for j in range(-3, 4):
            row = img[refl(y + j, h)]
            acc += int(gauss[j + 3]) * sum(int(gauss[i + 3]) * int(row[refl(x + i, w)]) for i in range(-3, 4))
        return (acc + (1 << 21)) >> 22

    near_border = 0
    for k, d in list(zip(kps, desc))[::7]:
        img = lv[k["octave"]]
        x, y = int(round(float(k["x"]) / float(scale[k["octave"]]))), int(round(float(k["y"]) / float(scale[k["octave"]])))
        m10 = sum(u * int(img[y + v, x + u]) for v in range(-15, 16) for u in range(-umax[abs(v)], umax[abs(v)] + 1))
        m01 = sum(v * int(img[y + v, x + u]) for v in range(-15, 16) for u in range(-umax[abs(v)], umax[abs(v)] + 1))
        ang = np.float32(oracle.orb_fast_atan2_deg(float(m01), float(m10)))
        assert ang == k["angle"]
        assert abs(float(ang) - np.degrees(np.arctan2(m01, m10)) % 360.0) < 0.02 or (m10 == 0 and m01 == 0)
        cs, sn = (np.float32(v) for v in oracle.orb_sincos_deg(ang))
        bits = np.zeros(256, np.uint8)
        for t in range(256):
            ax, ay, bx, by = (np.float32(v) for v in pat[t])
            ra = (int(np.rint(ax * cs - ay * sn)), int(np.rint(ax * sn + ay * cs)))
            rb = (int(np.rint(bx * cs - by * sn)), int(np.rint(bx * sn + by * cs)))
            bits[t] = blur(img, x + ra[0], y + ra[1]) < blur(img, x + rb[0], y + rb[1])
        assert np.array_equal(np.packbits(bits, bitorder="little"), d)
        h, w = img.shape
        near_border += int(min(x, y, w - 1 - x, h - 1 - y) < 22)
    assert near_border > 0  # some of the checked keypoints read mirrored pixels


# ---------------------------------------------------------------- quadtree distribution (oracle steps 4' and 5')
def _slam_candidates_py(S, ini_th):
    """Step 4' with array operations only (independent of the C loops): per-cell non-maximum suppression + threshold."""
    h, w = S.shape
    W2, H2 = w - 32, h - 32
    ncols, nrows = max(1, W2 // 30), max(1, H2 // 30)
    wc, hc = -(-W2 // ncols), -(-H2 // nrows)
    out = []
    for i in range(nrows):
        for j in range(ncols):
            x0, y0 = 19 + j * wc, 19 + i * hc
            x1, y1 = min(x0 + wc, w - 19), min(y0 + hc, h - 19)
            if x1 <= x0 or y1 <= y0:
                continue
            c = S[y0:y1, x0:x1].astype(np.int32)
            p = np.pad(c, 1)  # neighbours outside the cell's detection region count as 0
            nb = np.stack([p[1 + dy:1 + dy + c.shape[0], 1 + dx:1 + dx + c.shape[1]]
                           for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dy, dx) != (0, 0)]).max(axis=0)
            mx = (c > 0) & (c > nb)
            if (mx & (c > ini_th)).any():
                mx &= c > ini_th
            ys, xs = np.nonzero(mx)
            out += [(x0 + x, y0 + y, int(c[y, x])) for y, x in zip(ys, xs)]
    return out


def _quadtree_py(cands, w, h, N):
    """Step 5' as a level-synchronous set computation (no list, no pointers): a node is (x0, y0, x1, y1) -> [keys]."""
    W2, H2 = w - 32, h - 32
    f32 = np.float32
    n_ini = max(1, int(np.floor(float(f32(W2) / f32(H2)) + 0.5)))
    hX = f32(W2) / f32(n_ini)
    nodes = {}
    for (x, y, s) in cands:
        r = min(int(f32(x - 16) / hX), n_ini - 1)
        box = (int(hX * f32(r)), 0, int(hX * f32(r + 1)), H2)
        nodes.setdefault(box, []).append((x, y, s))

    def split(box, keys):
        x0, y0, x1, y1 = box
        xm, ym = x0 + (x1 - x0 + 1) // 2, y0 + (y1 - y0 + 1) // 2
        ch = {}
        for k in keys:
            left, up = k[0] - 16 < xm, k[1] - 16 < ym
            b = (x0 if left else xm, y0 if up else ym, xm if left else x1, ym if up else y1)
            ch.setdefault(b, []).append(k)
        return ch

    fresh = set(nodes)  # nodes created by the last pass
    while True:
        before = len(nodes)
        new = {}
        for box in [b for b in nodes if len(nodes[b]) > 1]:
            new.update(split(box, nodes.pop(box)))
        nodes.update(new)
        fresh = {b for b in new if len(new[b]) > 1}
        if len(nodes) >= N or len(nodes) == before:
            break
        if len(nodes) + 3 * len(fresh) > N:
            done = False
            while not done:
                before = len(nodes)
                order = sorted(fresh, key=lambda b: (-len(nodes[b]), b[1], b[0]))
                fresh = set()
                for b in order:
                    ch = split(b, nodes.pop(b))
                    nodes.update(ch)
                    fresh |= {c for c in ch if len(ch[c]) > 1}
                    if len(nodes) >= N:
                        break
                done = len(nodes) >= N or len(nodes) == before
            break
    win = [min(v, key=lambda k: (-k[2], k[1], k[0])) for v in nodes.values()]
    win = sorted(win, key=lambda k: (-k[2], k[1], k[0]))[:N]
    return sorted(win, key=lambda k: (k[1], k[0]))


def test_quadtree_distribution_against_an_independent_python_restatement(oracle):
    rng = np.random.default_rng(11)
    cases = [(oracle.synth_frame(640, 480, 3), 20, 7), (oracle.synth_frame(333, 257, 9), 20, 7),
             (rng.integers(0, 256, (200, 310), dtype=np.uint8), 60, 30),   # noise: dense candidates, deep trees
             (oracle.synth_frame(150, 420, 4), 20, 7)]                        # portrait: round(W'/H') = 0 -> one root
    for img, ini, mn in cases:
        h, w = img.shape
        S = oracle.orb_score_map(img, mn)
        cx, cy, cs = oracle.orb_slam_candidates(S, ini)
        assert sorted(zip(cx.tolist(), cy.tolist(), cs.tolist())) == sorted(_slam_candidates_py(S, ini))
        assert len(cx) > 50
        cands = list(zip(cx.tolist(), cy.tolist(), cs.tolist()))
        for N in (1, 2, 5, 17, 64, 200, 433, len(cands), len(cands) + 10):
            ox, oy, os_ = oracle.orb_quadtree(cx, cy, cs, w, h, N)
            assert list(zip(ox.tolist(), oy.tolist(), os_.tolist())) == _quadtree_py(cands, w, h, N), (w, h, N)
            assert len(ox) <= N
            if N >= len(cands):
                assert len(ox) == len(cands)  # enough room: every candidate ends alone in a node


def test_quadtree_distribution_spreads_the_keypoints(oracle):
    """What the mode is for: keypoints cover the image instead of piling up on the strongest texture."""
    img = oracle.synth_frame(640, 480, 21).copy()
    img[:, :320] = (img[:, :320].astype(np.int32) * 0.25 + 96).astype(np.uint8)  # left half: low contrast
    k0, _ = oracle.orb_extract(img, K=500)
    oracle.orb_set_distribution(1)
    try:
        k1, d1 = oracle.orb_extract(img, K=500)
        k1b, d1b = oracle.orb_extract(img, K=500)
    finally:
        oracle.orb_set_distribution(0)
    assert np.array_equal(k1.view(np.uint8), k1b.view(np.uint8)) and np.array_equal(d1, d1b)
    assert 0 < len(k1) <= 500

    def occupied(k):  # 80-px blocks of level-0 coordinates holding at least one level-0 keypoint
        m = k["octave"] == 0
        return len({(int(x) // 80, int(y) // 80) for x, y in zip(k["x"][m], k["y"][m])})
    assert occupied(k1) >= occupied(k0)
    q = oracle.orb_quotas(500)
    for l in range(8):
        assert (k1["octave"] == l).sum() <= q[l]
        m = k1["octave"] == l
        yx = np.stack([k1["y"][m], k1["x"][m]], 1)
        assert (np.lexsort((yx[:, 1], yx[:, 0])) == np.arange(len(yx))).all()  # (y, x) order inside a level
