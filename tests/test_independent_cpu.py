"""Independent CPU cross-checks of the two oracles the reference cannot pin (ORB, BA-LM): the reference ships no ORB
code and no optimizer plugin, so `oracle/orb_oracle.c` and `oracle/ba_oracle.c` are specified by this repo.  A GPU-vs-
oracle test cannot catch a misreading both share, so each oracle is checked here against a SECOND, naive restatement
written from the published definition, in numpy / scipy, sharing no code with the oracle or the kernels:

  BA   finite-difference Jacobians of the reprojection residual through scipy.linalg.expm (not the closed-form exp),
       and scipy.optimize.least_squares (loss='huber') reaching the same optimum cost as the oracle's LM;
  ORB  Rosten & Drummond's FAST-9 segment test (9 contiguous of 16 ring pixels all brighter / all darker by t), float
       bilinear pyramid, atan2 orientation, blur-whole-image-then-sample BRIEF.
"""
import ctypes as C
import math
import os
import sys

import numpy as np
import pytest
from scipy.linalg import expm
from scipy.optimize import least_squares

import oracle_lib
from gslam_amd.ba_synth import make_graph, quat_to_R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


# ------------------------------------------------------------------------------------------------ BA
def _hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])


def _pose_to_T(p):
    T = np.eye(4)
    T[:3, :3] = quat_to_R(np.asarray(p[:4])[None])[0]
    T[:3, 3] = p[4:]
    return T


def _retract_T(T, xi):
    """T * exp(xi), xi = [v, w] (GSLAM SE3.h:258-261 ordering), through the generic matrix exponential."""
    A = np.zeros((4, 4))
    A[:3, :3] = _hat(xi[3:])
    A[:3, 3] = xi[:3]
    return T @ expm(A)


def _project(T_wc, X):
    Xc = T_wc[:3, :3].T @ (X - T_wc[:3, 3])
    return Xc[:2] / Xc[2]


def _lin(oracle, pose, dof, X, pfree, m, info=None, huber=0.01):
    r, Jc, Jp = np.zeros(2), np.zeros(12), np.zeros(6)
    w, s = C.c_double(), C.c_double()
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    ok = oracle.lib.oracle_ba_obs_linearize(p(np.ascontiguousarray(pose)), int(dof), p(np.ascontiguousarray(X)), int(pfree),
                                            p(np.ascontiguousarray(m)), p(info), C.c_double(huber), p(r), C.byref(w),
                                            p(Jc), p(Jp), C.byref(s))
    return ok, r, w.value, Jc.reshape(2, 6), Jp.reshape(2, 3), s.value


def test_ba_jacobians_against_finite_differences(oracle):
    """d r / d xi (pose <- pose * exp(xi)) and d r / d X of the oracle's obs_linearize vs central differences of an
    independent projection (rotation matrix + scipy expm)."""
    rng = np.random.default_rng(7)
    g = make_graph(6, 40, n_obs_per_point=4, seed=13)
    h = 1e-6
    worst = 0.0
    for k in rng.choice(len(g["obs_cam"]), 40, replace=False):
        pose, X, m = g["cam_pose"][g["obs_cam"][k]], g["point_xyz"][g["obs_point"][k]], g["obs_xy"][k]
        ok, r, w, Jc, Jp, s = _lin(oracle, pose, 63, X, 1, m)
        assert ok == 1
        T = _pose_to_T(pose)
        assert np.allclose(r, _project(T, X) - m, atol=1e-13)
        Jc_fd = np.zeros((2, 6))
        for j in range(6):
            e = np.zeros(6)
            e[j] = h
            Jc_fd[:, j] = (_project(_retract_T(T, e), X) - _project(_retract_T(T, -e), X)) / (2 * h)
        Jp_fd = np.zeros((2, 3))
        for j in range(3):
            e = np.zeros(3)
            e[j] = h
            Jp_fd[:, j] = (_project(T, X + e) - _project(T, X - e)) / (2 * h)
        worst = max(worst, np.abs(Jc - Jc_fd).max() / max(1.0, np.abs(Jc).max()),
                    np.abs(Jp - Jp_fd).max() / max(1.0, np.abs(Jp).max()))
        # Huber IRLS weight and s
        assert abs(s - r @ r) < 1e-15
        assert abs(w - (1.0 if s <= 1e-4 else 0.01 / math.sqrt(s))) < 1e-15
    assert worst < 5e-8, worst


def test_ba_jacobian_dof_mask_and_information(oracle):
    g = make_graph(4, 10, n_obs_per_point=3, seed=2)
    pose, X, m = g["cam_pose"][1], g["point_xyz"][g["obs_point"][1]], g["obs_xy"][1]
    full = _lin(oracle, pose, 63, X, 1, m)
    masked = _lin(oracle, pose, 0b101010, X, 0, m)
    for j in range(6):
        assert np.array_equal(masked[3][:, j], full[3][:, j] if (0b101010 >> j) & 1 else np.zeros(2))
    assert np.array_equal(masked[4], np.zeros((2, 3)))
    info = np.array([2.0, 0.3, 0.3, 1.5])
    wi = _lin(oracle, pose, 63, X, 1, m, info=info, huber=0.0)
    assert abs(wi[5] - full[1] @ info.reshape(2, 2) @ full[1]) < 1e-15 and wi[2] == 1.0


def _scipy_problem(g, per_obs_norm):
    """Residual function of an independent minimiser: local pose parameters xi_c around g's poses (T_c * expm(xi_c)),
    points as they are; f = the 2-vectors r_k (plain least squares) or f_k = ||r_k|| (so that scipy's loss='huber',
    applied to f_k^2 = s_k with f_scale = delta, is the oracle's Huber on s_k)."""
    nc, npts = len(g["cam_pose"]), len(g["point_xyz"])
    T0 = [_pose_to_T(p) for p in g["cam_pose"]]
    dof = g["cam_dof"]
    free = [(c, j) for c in range(nc) for j in range(6) if (dof[c] >> j) & 1]
    ocam, opt, oxy = g["obs_cam"], g["obs_point"], g["obs_xy"]

    def fun(x):
        xi = np.zeros((nc, 6))
        for i, (c, j) in enumerate(free):
            xi[c, j] = x[i]
        P = x[len(free):].reshape(npts, 3)
        Tm = np.stack([_retract_T(T0[c], xi[c]) if dof[c] else T0[c] for c in range(nc)])
        Xc = np.einsum("nji,nj->ni", Tm[ocam, :3, :3], P[opt] - Tm[ocam, :3, 3])
        r = Xc[:, :2] / Xc[:, 2:3] - oxy
        return np.linalg.norm(r, axis=1) if per_obs_norm else r.ravel()

    return fun, np.concatenate([np.zeros(len(free)), g["point_xyz"].ravel()])


def _converged_options(huber):
    o = oracle_lib.ba_options(huber=huber, max_iterations=300)
    o.function_tolerance = 1e-15
    o.gradient_tolerance = 1e-14
    return o


@pytest.mark.parametrize("cams,pts,k,seed", [(4, 30, 4, 21), (5, 40, 5, 22), (6, 36, 6, 26)])
def test_ba_lm_reaches_the_optimum_minpack_finds(oracle, cams, pts, k, seed):
    """Plain least squares from the same start: MINPACK's Levenberg-Marquardt (scipy method='lm', numeric Jacobian of
    the independent residual) and the oracle's LM reach the same optimum cost to 1e-8 relative."""
    g = make_graph(cams, pts, n_obs_per_point=k, seed=seed, outlier_frac=0.05)
    _, _, s, rc = oracle.ba_solve(g, _converged_options(0.0))
    assert rc == 0 and s.termination in (1, 2)
    fun, x0 = _scipy_problem(g, per_obs_norm=False)
    res = least_squares(fun, x0, method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=20000)
    assert abs(s.final_cost - res.cost) <= 1e-8 * res.cost, (s.final_cost, res.cost)


@pytest.mark.parametrize("cams,pts,k,seed", [(4, 30, 4, 21), (5, 40, 5, 22), (6, 36, 6, 26)])
def test_ba_huber_optimum_is_stationary_for_scipy(oracle, cams, pts, k, seed):
    """Huber: scipy's trust-region-reflective with loss='huber' started AT the oracle's optimum must not find a lower
    cost (a wrong Jacobian sign or Huber weight in the oracle would leave a non-stationary point)."""
    g = make_graph(cams, pts, n_obs_per_point=k, seed=seed, outlier_frac=0.05)
    poses, pts_o, s, rc = oracle.ba_solve(g, _converged_options(0.01))
    assert rc == 0 and s.termination in (1, 2) and s.final_cost < 0.8 * s.initial_cost
    g2 = dict(g, cam_pose=poses, point_xyz=pts_o)
    fun, x0 = _scipy_problem(g2, per_obs_norm=True)
    f0 = fun(x0)
    assert abs(0.5 * np.where(f0 > 0.01, 2 * 0.01 * f0 - 1e-4, f0 * f0).sum() - s.final_cost) <= 1e-12 * s.final_cost
    res = least_squares(fun, x0, loss="huber", f_scale=0.01, method="trf", jac="3-point", xtol=1e-15, ftol=1e-15,
                        gtol=1e-14, max_nfev=300)
    assert abs(s.final_cost - res.cost) <= 1e-8 * res.cost, (s.final_cost, res.cost)
    assert res.cost >= s.final_cost * (1 - 1e-9)


def test_ba_huber_optimum_recovered_by_scipy_from_a_perturbed_start(oracle):
    """...and from a start 1e-4 away scipy comes back to the same Huber cost to 1e-8 (local minimum, not a saddle)."""
    g = make_graph(4, 30, n_obs_per_point=4, seed=21, outlier_frac=0.05)
    poses, pts_o, s, rc = oracle.ba_solve(g, _converged_options(0.01))
    fun, x0 = _scipy_problem(dict(g, cam_pose=poses, point_xyz=pts_o), per_obs_norm=True)
    x1 = x0 + np.random.default_rng(1).standard_normal(len(x0)) * 1e-4
    res = least_squares(fun, x1, loss="huber", f_scale=0.01, method="trf", jac="3-point", xtol=1e-15, ftol=1e-15,
                        gtol=1e-14, max_nfev=400)
    assert abs(s.final_cost - res.cost) <= 1e-8 * res.cost, (s.final_cost, res.cost)


def test_pnp_oracle_against_scipy(oracle):
    """oracle_ba_pnp (motion-only BA, Optimizer.h:202-207) on noisy matches with outliers vs scipy's Huber optimum."""
    g = make_graph(3, 120, n_obs_per_point=3, seed=31, noise=0.0, outlier_frac=0.0, perturb=False)
    sel = g["obs_cam"] == 1
    X = g["point_xyz_gt"][g["obs_point"][sel]]
    rng = np.random.default_rng(5)
    m = g["obs_xy"][sel] + rng.standard_normal((sel.sum(), 2)) * 0.002
    m[::9] += rng.standard_normal((len(m[::9]), 2)) * 0.1
    start = oracle.se3_retract(g["cam_pose_gt"][1], np.array([0.05, -0.04, 0.03, 0.01, -0.02, 0.015]))
    o = oracle_lib.ba_options(huber=0.01, max_iterations=100)
    o.function_tolerance = 1e-15
    o.gradient_tolerance = 1e-14
    pose, s, info, rc = oracle.ba_pnp(X, m, start, opts=o, want_information=True)
    assert rc == 0 and s.final_cost < s.initial_cost
    gg = {"cam_pose": start[None], "cam_dof": np.array([63], np.int32), "point_xyz": X,
          "obs_cam": np.zeros(len(X), np.int32), "obs_point": np.arange(len(X), dtype=np.int32), "obs_xy": m}
    T0 = _pose_to_T(start)

    def fun(x):
        T = _retract_T(T0, x)
        return np.array([np.linalg.norm(_project(T, X[k]) - m[k]) for k in range(len(X))])

    res = least_squares(fun, np.zeros(6), loss="huber", f_scale=0.01, jac="3-point", xtol=1e-15, ftol=1e-15, gtol=1e-14)
    assert abs(s.final_cost - res.cost) <= 1e-8 * res.cost
    assert abs(oracle.ba_cost(gg, pose[None], X) - s.final_cost) <= 1e-12 * s.final_cost
    assert np.allclose(info, info.T) and np.all(np.linalg.eigvalsh(info) > 0)


# ------------------------------------------------------------------------------------------------ ORB
# Bresenham circle of radius 3, clockwise from 12 o'clock (Rosten & Drummond 2006, fig. 1) -- written down here
# independently of include/gslam_orb_tables.h
RING16 = [(0, -3), (1, -3), (2, -2), (3, -1), (3, 0), (3, 1), (2, 2), (1, 3),
          (0, 3), (-1, 3), (-2, 2), (-3, 1), (-3, 0), (-3, -1), (-2, -2), (-1, -3)]


def _naive_fast9_count(img):
    """For every interior pixel: the number of thresholds t = 0, 1, 2, ... at which the pixel passes the segment test
    'there are 9 contiguous ring pixels all > p + t, or all < p - t'.  The test is monotone in t, so this count is the
    largest passing threshold + 1 == the oracle's score (the largest t for which the pixel is still a corner is
    score - 1, i.e. corner at t  <=>  score > t)."""
    h, w = img.shape
    I = img.astype(np.int32)
    c = I[3:h - 3, 3:w - 3]
    ring = np.stack([I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in RING16])  # 16 x H x W
    count = np.zeros_like(c)
    for t in range(0, 255):
        br = ring > c + t
        dk = ring < c - t
        is_corner = np.zeros(c.shape, bool)
        for a in range(16):
            idx = [(a + i) % 16 for i in range(9)]
            is_corner |= br[idx].all(axis=0) | dk[idx].all(axis=0)
        if not is_corner.any():
            break
        count += is_corner
    return count


def test_fast9_score_against_naive_segment_test(oracle):
    rng = np.random.default_rng(11)
    imgs = [oracle.synth_frame(160, 120, 0x5EED0000 + 3),
            rng.integers(0, 256, (64, 80), dtype=np.uint8),
            (rng.integers(0, 2, (70, 70)) * 200 + rng.integers(0, 40, (70, 70))).astype(np.uint8)]
    # smooth random blobs: many genuine corners with mid-range scores
    yy, xx = np.mgrid[0:90, 0:110]
    blob = np.zeros((90, 110))
    for _ in range(25):
        cx, cy, r, v = rng.uniform(0, 110), rng.uniform(0, 90), rng.uniform(4, 15), rng.uniform(-120, 120)
        blob += v * ((xx - cx) ** 2 + (yy - cy) ** 2 < r * r)
    imgs.append(np.clip(blob + 128, 0, 255).astype(np.uint8))
    n_corners = 0
    for img in imgs:
        h, w = img.shape
        for min_th in (0, 7, 20):
            S = oracle.orb_score_map(img, min_th=min_th).astype(np.int32)
            naive = _naive_fast9_count(img)
            full = np.zeros((h, w), np.int32)
            full[3:h - 3, 3:w - 3] = naive
            valid = np.zeros((h, w), bool)
            valid[19:h - 19, 19:w - 19] = True  # the oracle scores only the 19-px-inset region (spec step 2)
            want = np.where(valid & (full > min_th), np.minimum(full, 255), 0)
            assert np.array_equal(S, want), f"{img.shape} min_th={min_th}: {np.argwhere(S != want)[:5]}"
            n_corners += int((want > 0).sum())
    assert n_corners > 500


def test_pyramid_against_float_bilinear(oracle):
    """11-bit fixed-point bilinear 1.2x chain vs float64 bilinear at the pixel-centre mapping: within 1 grey level
    per level (weights are quantised to 1/2048, result rounded half up)."""
    g = oracle.synth_frame(333, 257, 5)
    ws, hs = oracle.orb_level_dims(333, 257)
    for l in range(1, 8):
        assert ws[l] == int(math.floor(333 * (5 / 6) ** l + 0.5)) and hs[l] == int(math.floor(257 * (5 / 6) ** l + 0.5))
    prev = g
    for l in range(1, 4):
        cur = oracle.orb_pyramid_level(g, l)
        hd, wd = cur.shape
        hs_, ws_ = prev.shape
        sx = np.clip((np.arange(wd) + 0.5) * ws_ / wd - 0.5, 0, ws_ - 1)
        sy = np.clip((np.arange(hd) + 0.5) * hs_ / hd - 0.5, 0, hs_ - 1)
        x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
        x1, y1 = np.minimum(x0 + 1, ws_ - 1), np.minimum(y0 + 1, hs_ - 1)
        fx, fy = (sx - x0)[None, :], (sy - y0)[:, None]
        P = prev.astype(np.float64)
        ref = (P[y0][:, x0] * (1 - fx) + P[y0][:, x1] * fx) * (1 - fy) + (P[y1][:, x0] * (1 - fx) + P[y1][:, x1] * fx) * fy
        assert np.abs(cur.astype(np.float64) - ref).max() <= 1.0
        assert np.abs(cur.astype(np.float64) - ref).mean() < 0.3
        prev = cur


def _naive_blur(img):
    """7x7 sigma = 2 Gaussian with integer taps summing to 2048 per axis, applied to the WHOLE image, then
    (acc + 2^21) >> 22 (spec step 7)."""
    w = np.array([math.exp(-(i * i) / 8.0) for i in range(-3, 4)])
    taps = np.rint(2048 * w / w.sum()).astype(np.int64)
    taps[3] += 2048 - taps.sum()
    I = img.astype(np.int64)
    h, wd = I.shape
    hp = np.zeros_like(I)
    for i in range(7):
        hp[:, 3:wd - 3] += taps[i] * I[:, i:wd - 6 + i]
    out = np.zeros_like(I)
    for j in range(7):
        out[3:h - 3, :] += taps[j] * hp[j:h - 6 + j, :]
    return (out + (1 << 21)) >> 22


def test_descriptor_and_orientation_against_naive_restatement(oracle):
    """Level-0 keypoints of the oracle: orientation bin from float atan2 of the intensity-centroid moments over the ORB
    disc, descriptor from blur-the-whole-image-then-sample with the pattern rotated by the bin's angle in float and
    rounded -- all recomputed here from the published definitions (Rublee et al. 2011, sections 3.2 and 4.2)."""
    import gen_orb_tables as T  # the seeded generator of the spec data (tools/), not the C header
    pattern = T.gen_pattern()
    img = oracle.synth_frame(320, 240, 0x5EED0000 + 1)
    kps, desc = oracle.orb_extract(img, 300, nlevels=1)
    assert len(kps) > 200
    B = _naive_blur(img)
    I = img.astype(np.int64)
    # ORB's disc: |u| <= floor(sqrt(15^2 - v^2) + 0.5) for |v| <= 10, mirrored so that the patch is symmetric
    # (OpenCV's u_max construction); built by the generator from that rule
    umax = T.gen_umax()
    checked = 0
    for kp, d in zip(kps, desc):
        x, y = int(kp["x"]), int(kp["y"])
        m10 = m01 = 0
        for v in range(-15, 16):
            um = umax[abs(v)]
            row = I[y + v, x - um:x + um + 1]
            m10 += int((np.arange(-um, um + 1) * row).sum())
            m01 += int(v * row.sum())
        ang = math.degrees(math.atan2(m01, m10)) % 360.0
        k = int(math.floor((ang + 6.0) / 12.0)) % 30
        frac = ((ang + 6.0) / 12.0) % 1.0
        if min(frac, 1 - frac) < 1e-3:
            continue  # within 0.012 degrees of a bin boundary: the integer boundary table may round the other way
        assert kp["angle"] == 12.0 * k, (x, y, ang, kp["angle"])
        th = math.radians(12.0 * k)
        c, s = math.cos(th), math.sin(th)
        bits = np.zeros(256, np.uint8)
        for i, (ax, ay, bx, by) in enumerate(pattern):
            rx = lambda v: int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)
            a = B[y + rx(ax * s + ay * c), x + rx(ax * c - ay * s)]
            b = B[y + rx(bx * s + by * c), x + rx(bx * c - by * s)]
            bits[i] = a < b
        assert np.array_equal(np.packbits(bits, bitorder="little"), d), (x, y)
        checked += 1
    assert checked > 200


def test_selected_keypoints_are_strict_local_maxima_of_the_naive_score(oracle):
    img = oracle.synth_frame(320, 240, 0x5EED0000 + 2)
    kps, _ = oracle.orb_extract(img, 400, nlevels=1)
    h, w = img.shape
    full = np.zeros((h, w), np.int32)
    full[3:h - 3, 3:w - 3] = _naive_fast9_count(img)
    valid = np.zeros((h, w), bool)
    valid[19:h - 19, 19:w - 19] = True
    S = np.where(valid & (full > 7), full, 0)
    for kp in kps:
        x, y = int(kp["x"]), int(kp["y"])
        assert S[y, x] == int(kp["response"]) > 7
        nb = S[y - 1:y + 2, x - 1:x + 2].copy()
        nb[1, 1] = -1
        assert nb.max() < S[y, x]
    # per 32x32 cell anchored at (19,19): if the cell holds a local maximum above 20, nothing <= 20 was kept from it
    cx = (kps["x"].astype(int) - 19) // 32
    cy = (kps["y"].astype(int) - 19) // 32
    for c in set(zip(cx.tolist(), cy.tolist())):
        r = kps["response"][(cx == c[0]) & (cy == c[1])]
        assert (r > 20).all() or (r <= 20).all()


def _naive_select_level0(img, K, ini_th=20, min_th=7):
    """Spec steps 2-5 for a one-level pyramid, written from the numbered specification in plain Python (no code shared
    with oracle/orb_oracle.c or the kernels): naive FAST-9 score, strict 3x3 NMS, 32x32 cells anchored at (19, 19)
    with the strong-corner rule, in-cell rank by (S desc, y, x) capped at 32, quota K under the order
    (rank, S desc, cell, raster), output in (cell, raster) order.  Returns [(x, y, S)]."""
    h, w = img.shape
    full = np.zeros((h, w), np.int32)
    full[3:h - 3, 3:w - 3] = np.minimum(_naive_fast9_count(img), 255)
    S = np.zeros((h, w), np.int32)
    S[19:h - 19, 19:w - 19] = full[19:h - 19, 19:w - 19]
    S[S <= min_th] = 0
    ncx = (w - 38 + 31) // 32
    cells = {}
    for y in range(19, h - 19):
        for x in range(19, w - 19):
            s = S[y, x]
            if s == 0:
                continue
            nb = S[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if nb.max() < s:
                cells.setdefault(((y - 19) // 32) * ncx + (x - 19) // 32, []).append((int(s), y, x))
    ranked = []
    for c, lst in cells.items():
        if any(s > ini_th for s, _, _ in lst):
            lst = [e for e in lst if e[0] > ini_th]
        lst.sort(key=lambda e: (-e[0], e[1], e[2]))
        ranked += [(r, -s, c, y, x) for r, (s, y, x) in enumerate(lst[:32])]
    ranked.sort()
    take = ranked[:K]
    take.sort(key=lambda e: (e[2], e[3], e[4]))
    return [(x, y, -ns) for _, ns, _, y, x in take]


@pytest.mark.parametrize("name", ["noise", "binary_noise", "low_contrast", "dots8", "checker2", "checker5", "mixed",
                                  "step_edges", "sparse_binary"])
def test_selection_against_naive_restatement_on_adversarial_images(oracle, name):
    """Saturated cells, the 32-entry cap, quota cuts through tied scores: the oracle's one-level selection (positions,
    responses AND output order) equals the plain-Python restatement of spec steps 2-5."""
    from orb_images import CLASSES
    img = CLASSES[name](150, 131, 5)
    n_total = 0
    for K, ini in ((40, 20), (700, 20), (6000, 35)):
        kps, _ = oracle.orb_extract(img, K, nlevels=1, ini_th=ini, min_th=7)
        want = _naive_select_level0(img, K, ini_th=ini)
        got = [(int(k["x"]), int(k["y"]), int(k["response"])) for k in kps]
        assert got == want, (name, K, len(got), len(want))
        n_total += len(got)
    assert n_total > 0 or name == "checker2"  # a 2-px checkerboard has no 9-arc at level 0: both sides agree on "nothing"
