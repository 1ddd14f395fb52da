"""CPU checks of the RANSAC oracle against ground-truth models on synthetic correspondences (parity unpinned: the
reference has only the Estimator interface)."""
import numpy as np


def _corr(model, n, outlier_frac, seed, noise):
    rng = np.random.default_rng(seed)
    if model == 3:
        A = rng.normal(size=(3, 4))
        P = rng.uniform(-5, 5, (n, 3))
        Q = P @ A[:, :3].T + A[:, 3] + rng.normal(size=(n, 3)) * noise
        truth = A.reshape(-1)
    else:
        P = rng.uniform(0, 640, (n, 2))
        if model == 0:
            H = np.array([[1.1, 0.05, 20], [-0.04, 0.95, -10], [1e-4, -5e-5, 1.0]])
            ph = np.c_[P, np.ones(n)] @ H.T
            Q = ph[:, :2] / ph[:, 2:3]
            truth = (H / H[2, 2]).reshape(-1)
        elif model == 1:
            A = np.array([[0.9, -0.2, 15.0], [0.25, 1.05, -7.0]])
            Q = np.c_[P, np.ones(n)] @ A.T
            truth = A.reshape(-1)
        else:  # two views of 3D points
            X = np.c_[rng.uniform(-3, 3, (n, 2)), rng.uniform(4, 9, n)]
            K = np.array([[500.0, 0, 320], [0, 500, 240], [0, 0, 1]])
            th = 0.1
            R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
            t = np.array([0.5, 0.05, 0.1])
            p1 = X @ K.T
            P = p1[:, :2] / p1[:, 2:3]
            p2 = (X @ R.T + t) @ K.T
            Q = p2[:, :2] / p2[:, 2:3]
            truth = None
        Q = Q + rng.normal(size=Q.shape) * noise
    out = rng.random(n) < outlier_frac
    Q[out] += rng.uniform(30, 200, (int(out.sum()), Q.shape[1])) * rng.choice([-1, 1], (int(out.sum()), Q.shape[1]))
    return P, Q, ~out, truth


def test_homography_affine_recover_truth_and_mask(oracle):
    for model, thr in ((0, 2.0), (1, 2.0), (3, 0.05)):
        P, Q, inl, truth = _corr(model, 600, 0.3, 10 + model, 0.3 if model != 3 else 0.005)
        m, mask, cnt = oracle.ransac(model, P, Q, thr)
        assert cnt == mask.sum() and cnt >= 0.95 * inl.sum()
        assert (mask.astype(bool) & ~inl).sum() <= 3  # outliers rejected
        ms = len(truth)
        assert np.allclose(m[:ms], truth, rtol=0.08, atol=0.6)


def test_fundamental_epipolar_constraint(oracle):
    P, Q, inl, _ = _corr(2, 800, 0.25, 21, 0.2)
    m, mask, cnt = oracle.ransac(2, P, Q, 1.0)
    F = m[:9].reshape(3, 3)
    assert cnt >= 0.9 * inl.sum() and (mask.astype(bool) & ~inl).mean() < 0.03
    x1 = np.c_[P[inl], np.ones(inl.sum())]
    x2 = np.c_[Q[inl], np.ones(inl.sum())]
    l = x1 @ F.T
    d = np.abs(np.sum(x2 * l, axis=1)) / np.hypot(l[:, 0], l[:, 1])
    assert np.median(d) < 1.0  # point-to-epipolar-line distance of the true inliers


def test_degenerate_and_small_inputs(oracle):
    P = np.zeros((10, 2))
    m, mask, cnt = oracle.ransac(0, P, P, 1.0)  # all points identical: every sample is singular
    assert cnt == 0 and not mask.any() and not m.any()
    m, mask, cnt = oracle.ransac(0, np.zeros((3, 2)), np.zeros((3, 2)), 1.0)  # fewer than the sample size
    assert cnt == 0
    rng = np.random.default_rng(0)
    P = rng.uniform(0, 100, (4, 2))
    m, mask, cnt = oracle.ransac(1, P, P * 2 + 1, 1e-6)
    assert cnt == 4 and np.allclose(m[:6], [2, 0, 1, 0, 2, 1], atol=1e-9)


def test_deterministic_in_seed(oracle):
    P, Q, _, _ = _corr(0, 300, 0.4, 5, 0.5)
    a = oracle.ransac(0, P, Q, 2.0, seed=7)
    b = oracle.ransac(0, P, Q, 2.0, seed=7)
    c = oracle.ransac(0, P, Q, 2.0, seed=8)
    assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1])
    assert a[0].tobytes() != c[0].tobytes()


# ---------------------------------------------------------------- essential / SIM3 / plane / PnP / triangulation
def _rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def _two_view(n, outlier_frac, seed, noise):
    """Normalised image coordinates of n points in two views; returns p1, p2, inlier flags, (R, t) with X2 = R X1 + t."""
    rng = np.random.default_rng(seed)
    X = np.c_[rng.uniform(-3, 3, (n, 2)), rng.uniform(4, 9, n)]
    R, t = _rot([0.2, 1.0, 0.1], 0.12), np.array([0.6, 0.05, 0.1])
    X2 = X @ R.T + t
    p1 = X[:, :2] / X[:, 2:3]
    p2 = X2[:, :2] / X2[:, 2:3] + rng.normal(size=(n, 2)) * noise
    out = rng.random(n) < outlier_frac
    p2[out] += rng.uniform(0.05, 0.3, (int(out.sum()), 2)) * rng.choice([-1, 1], (int(out.sum()), 2))
    return p1, p2, ~out, (R, t)


def test_essential_has_two_equal_singular_values_and_fits_the_geometry(oracle):
    p1, p2, inl, (R, t) = _two_view(700, 0.25, 31, 0.0005)
    m, mask, cnt = oracle.ransac(4, p1, p2, 0.002)
    E = m[:9].reshape(3, 3)
    sv = np.linalg.svd(E, compute_uv=False)
    assert abs(sv[0] - sv[1]) <= 1e-9 * sv[0] and sv[2] <= 1e-9 * sv[0]
    assert cnt >= 0.9 * inl.sum() and (mask.astype(bool) & ~inl).sum() <= 5
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Et = tx @ R
    cosang = abs(np.sum(E * Et)) / (np.linalg.norm(E) * np.linalg.norm(Et))
    assert cosang > 0.999  # the same matrix up to scale and sign


def test_sim3_plane_pnp_recover_ground_truth(oracle):
    rng = np.random.default_rng(7)
    n = 500
    # SIM3
    A = rng.uniform(-4, 4, (n, 3))
    R, t, s = _rot([1, 2, 3], 0.7), np.array([1.0, -2.0, 0.5]), 1.7
    B = s * A @ R.T + t + rng.normal(size=(n, 3)) * 0.002
    out = rng.random(n) < 0.3
    B[out] += rng.uniform(0.5, 2, (int(out.sum()), 3))
    m, mask, cnt = oracle.ransac(5, A, B, 0.02)
    q = m[:4]
    Rm = np.array([[1 - 2 * (q[1] ** 2 + q[2] ** 2), 2 * (q[0] * q[1] - q[3] * q[2]), 2 * (q[0] * q[2] + q[3] * q[1])],
                   [2 * (q[0] * q[1] + q[3] * q[2]), 1 - 2 * (q[0] ** 2 + q[2] ** 2), 2 * (q[1] * q[2] - q[3] * q[0])],
                   [2 * (q[0] * q[2] - q[3] * q[1]), 2 * (q[1] * q[2] + q[3] * q[0]), 1 - 2 * (q[0] ** 2 + q[1] ** 2)]])
    assert np.abs(Rm - R).max() < 5e-3 and abs(m[7] - s) < 5e-3 and np.abs(m[4:7] - t).max() < 2e-2
    assert cnt >= 0.9 * (~out).sum() and (mask.astype(bool) & out).sum() <= 3
    # plane
    nrm = np.array([0.2, -0.3, 0.93]); nrm /= np.linalg.norm(nrm)
    P = rng.uniform(-5, 5, (n, 3))
    P -= np.outer(P @ nrm + 1.5, nrm)  # on the plane n . x + 1.5 = 0
    P += np.outer(rng.normal(size=n) * 0.002, nrm)
    off = rng.random(n) < 0.3
    P[off] += np.outer(rng.uniform(0.2, 2, int(off.sum())) * rng.choice([-1, 1], int(off.sum())), nrm)
    m, mask, cnt = oracle.ransac(6, P, P, 0.01)
    sgn = np.sign(m[:3] @ nrm)
    assert np.abs(sgn * m[:3] - nrm).max() < 2e-3 and abs(sgn * m[3] - 1.5) < 5e-3
    assert cnt >= 0.95 * (~off).sum() and (mask.astype(bool) & off).sum() == 0
    # PnP (non-planar object points)
    X = np.c_[rng.uniform(-3, 3, (n, 2)), rng.uniform(-1, 1, n)]
    Rc, tc = _rot([0.3, -1, 0.2], 0.4), np.array([0.2, -0.1, 6.0])
    Xc = X @ Rc.T + tc
    uv = Xc[:, :2] / Xc[:, 2:3] + rng.normal(size=(n, 2)) * 0.0003
    bad = rng.random(n) < 0.3
    uv[bad] += rng.uniform(0.03, 0.2, (int(bad.sum()), 2))
    m, mask, cnt = oracle.ransac(7, X, uv, 0.003)
    Rm = m[:9].reshape(3, 3)
    assert np.abs(Rm @ Rm.T - np.eye(3)).max() < 1e-12 and np.linalg.det(Rm) > 0.999
    assert np.abs(Rm - Rc).max() < 2e-2 and np.abs(m[9:12] - tc).max() < 0.1
    assert cnt >= 0.8 * (~bad).sum() and (mask.astype(bool) & bad).sum() <= 3


def test_triangulation_midpoint(oracle):
    rng = np.random.default_rng(3)
    R, t = _rot([0.1, 1, 0.2], 0.2), np.array([-0.8, 0.02, 0.05])
    from gslam_amd.ba_synth import _quat_from_R
    pose = np.r_[_quat_from_R(R[None])[0], t]
    for _ in range(50):
        X = np.r_[rng.uniform(-2, 2, 2), rng.uniform(3, 10)]
        X2 = R @ X + t
        got, ok = oracle.triangulate(pose, X / X[2], X2 / X2[2])
        assert ok and np.abs(got - X).max() < 1e-9
    _, ok = oracle.triangulate(pose, np.array([0, 0, 1.0]), R @ np.array([0, 0, 1.0]))  # parallel rays
    assert not ok
    _, ok = oracle.triangulate(pose, np.array([-0.5, 0, 1.0]), np.array([0.9, 0, 1.0]))  # diverging: behind a camera
    assert not ok


def test_confidence_is_the_sequential_adaptive_stopping_rule(oracle):
    """Estimator.h:100-169 pass threshold AND confidence.  oracle_ransac_conf restates sequential RANSAC with the textbook
    stopping rule; re-derive it here from oracle_ransac's own hypothesis stream: scoring prefixes of the hypothesis list
    is the same as running the full search with the data the prefix saw, so the winner at confidence c must be the best of
    the first `used` hypotheses and `used` must satisfy the rule for that winner but for no earlier best."""
    import math
    for model, thr, s in ((0, 2.0, 4), (1, 2.0, 3), (2, 1.0, 8)):
        P, Q, inl, _ = _corr(model, 600, 0.35, 900 + model, 0.3)
        m_full, mask_full, cnt_full = oracle.ransac(model, P, Q, thr, seed=5)
        m1, mask1, cnt1, used1 = oracle.ransac_conf(model, P, Q, thr, 1.0, seed=5)
        assert used1 == 2048 and cnt1 == cnt_full and m1.tobytes() == m_full.tobytes()
        prev_used = 0
        for conf in (0.5, 0.9, 0.99, 0.999999):
            m, mask, cnt, used = oracle.ransac_conf(model, P, Q, thr, conf, seed=5)
            assert 1 <= used <= 2048 and used >= prev_used and 0 < cnt <= cnt_full
            prev_used = used
            w = cnt / len(P)
            need = math.ceil(math.log(1 - conf) / math.log(1 - w ** s)) if 0 < w < 1 else 1
            assert used >= min(max(need, 1), 2048)  # the walk never stops before the rule holds for its winner
            assert mask.sum() == cnt
        assert prev_used < 2048 or cnt_full / len(P) < 0.3  # a 65 % inlier set never needs all 2048 draws of 4 / 3 points


# ---------------------------------------------------------------- LMEDS and NOSAMPLE (GSLAM/core/Estimator.h:86-89)
def _all_model_cases():
    """(model, src, dst, threshold, inlier flags) for the eight models, 25-30 % outliers."""
    out = []
    for model, thr, noise in ((0, 2.0, 0.3), (1, 2.0, 0.3), (3, 0.05, 0.005), (2, 1.0, 0.2)):
        P, Q, inl, _ = _corr(model, 600, 0.28, 40 + model, noise)
        out.append((model, P, Q, thr, inl))
    p1, p2, inl, _ = _two_view(600, 0.25, 33, 0.0005)
    out.append((4, p1, p2, 0.002, inl))
    rng = np.random.default_rng(17)
    n = 600
    bad = rng.random(n) < 0.3
    A = rng.uniform(-4, 4, (n, 3))
    B = 1.7 * A @ _rot([1, 2, 3], 0.7).T + np.array([1.0, -2.0, 0.5]) + rng.normal(size=(n, 3)) * 0.002
    B[bad] += rng.uniform(0.5, 2, (int(bad.sum()), 3))
    out.append((5, A, B, 0.02, ~bad))
    nrm = np.array([0.2, -0.3, 0.93]); nrm /= np.linalg.norm(nrm)
    P = rng.uniform(-5, 5, (n, 3))
    P -= np.outer(P @ nrm + 1.5, nrm)
    P += np.outer(rng.normal(size=n) * 0.002, nrm)
    P[bad] += np.outer(rng.uniform(0.2, 2, int(bad.sum())) * rng.choice([-1, 1], int(bad.sum())), nrm)
    out.append((6, P, P, 0.01, ~bad))
    X = np.c_[rng.uniform(-3, 3, (n, 2)), rng.uniform(-1, 1, n)]
    Xc = X @ _rot([0.3, -1, 0.2], 0.4).T + np.array([0.2, -0.1, 6.0])
    uv = Xc[:, :2] / Xc[:, 2:3] + rng.normal(size=(n, 2)) * 0.0003
    uv[bad] += rng.uniform(0.03, 0.2, (int(bad.sum()), 2))
    out.append((7, X, uv, 0.003, ~bad))
    return out


def test_lmeds_needs_no_threshold_and_finds_the_inliers(oracle):
    """Least median of squares: with under half of the correspondences wrong, the hypothesis of smallest median error is a
    clean one and the robust sigma rule separates the two populations -- with threshold 0 (none given)."""
    for model, P, Q, thr, inl in _all_model_cases():
        m, mask, cnt, used = oracle.estimate_ex(model, P, Q, 0.0, 1)
        assert used == 2048 and cnt == mask.sum()
        assert cnt >= 0.85 * inl.sum(), (model, cnt, inl.sum())
        assert (mask.astype(bool) & ~inl).sum() <= 0.03 * len(inl), model
        # a threshold above the robust sigma widens the mask and never shrinks it
        m2, mask2, cnt2, _ = oracle.estimate_ex(model, P, Q, 10 * thr, 1)
        assert m2.tobytes() == m.tobytes() and cnt2 >= cnt and not (mask.astype(bool) & ~mask2.astype(bool)).any()
    # more than half outliers: the median is an outlier error, LMedS is not the tool (RANSAC still is)
    P, Q, inl, _ = _corr(0, 600, 0.6, 5, 0.3)
    _, mask, cnt, _ = oracle.estimate_ex(0, P, Q, 0.0, 1)
    _, rmask, rcnt = oracle.ransac(0, P, Q, 2.0)
    assert rcnt >= 0.9 * inl.sum()


def test_nosample_is_the_least_squares_fit_of_all_points(oracle):
    """NOSAMPLE on clean data (noise, no outliers) recovers the generating model at least as well as a minimal sample; on the
    exact minimal number of points it reproduces the minimal solver's model; outliers pull it away (that is what it is)."""
    clean = []
    for model, thr, noise in ((0, 2.0, 0.3), (1, 2.0, 0.3), (3, 0.05, 0.005)):
        P, Q, inl, truth = _corr(model, 400, 0.0, 70 + model, noise)
        m, mask, cnt, used = oracle.estimate_ex(model, P, Q, thr, 2)
        assert used == 1 and cnt == mask.sum() and cnt >= 0.99 * len(P)
        assert np.allclose(m[:len(truth)], truth, rtol=2e-3, atol=0.15), model
        rm, _, _ = oracle.ransac(model, P, Q, thr)
        assert np.abs(m[:len(truth)] - truth).max() <= np.abs(rm[:len(truth)] - truth).max() + 1e-12
        clean.append((model, P, Q, thr))
    for model, P, Q, thr, inl in _all_model_cases():
        Pi, Qi = P[inl], Q[inl]
        m, mask, cnt, used = oracle.estimate_ex(model, Pi, Qi, thr, 2)
        assert used == 1 and cnt >= 0.9 * len(Pi), (model, cnt, len(Pi))
        mo, masko, cnto, _ = oracle.estimate_ex(model, P, Q, thr, 2)  # with the outliers in: a worse fit
        assert cnto < cnt
    # exactly the minimal sample: the same model as the minimal solver up to rounding (H: 4 pairs)
    rng = np.random.default_rng(1)
    P = rng.uniform(0, 100, (4, 2))
    H = np.array([[1.05, 0.02, 3.0], [-0.03, 0.97, -2.0], [1e-4, 2e-4, 1.0]])
    ph = np.c_[P, np.ones(4)] @ H.T
    Q = ph[:, :2] / ph[:, 2:3]
    m, mask, cnt, _ = oracle.estimate_ex(0, P, Q, 1e-6, 2)
    assert cnt == 4 and np.allclose(m[:9], H.reshape(-1), atol=1e-7)
    m, _, cnt, used = oracle.estimate_ex(0, P[:3], Q[:3], 1.0, 2)  # fewer points than unknowns: no model
    assert cnt == 0 and used == 0 and not m.any()
