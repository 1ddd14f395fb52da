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
