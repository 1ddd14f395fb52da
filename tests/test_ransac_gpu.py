"""GPU parity of the RANSAC estimator (HIP, through the C ABI) vs the oracle: the winning model's doubles and the
inlier mask must be bit-identical (same sampling, same IEEE operation order, no FMA contraction)."""
import numpy as np
import pytest

from test_ransac_oracle import _corr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model,thr,n,noise", [(0, 2.0, 2000, 0.4), (1, 2.0, 500, 0.3), (2, 1.0, 2000, 0.3),
                                               (3, 0.05, 700, 0.005), (0, 3.0, 9, 0.1), (2, 1.5, 8, 0.0)])
def test_ransac_bit_exact_vs_oracle(ctx, oracle, model, thr, n, noise):
    from gslam_amd import estimator
    P, Q, inl, _ = _corr(model, n, 0.3, 100 + model + n, noise)
    for seed in (1, 12345):
        em, emask, ecnt = oracle.ransac(model, P, Q, thr, seed=seed)
        gm, gmask, gcnt = estimator.estimate(ctx, model, P, Q, thr, seed=seed)
        assert gcnt == ecnt and np.array_equal(gmask, emask)
        assert gm.tobytes() == em.tobytes()
    assert ecnt >= 0.8 * inl.sum() or n < 20


def test_ransac_degenerate(ctx, oracle):
    from gslam_amd import estimator
    P = np.zeros((50, 2))
    m, mask, cnt = estimator.estimate(ctx, 0, P, P, 1.0)
    assert cnt == 0 and not mask.any() and not m.any()
    m, mask, cnt = estimator.estimate(ctx, 2, np.zeros((5, 2)), np.zeros((5, 2)), 1.0)
    assert cnt == 0


def _cases():
    from test_ransac_oracle import _rot, _two_view
    rng = np.random.default_rng(77)
    n = 800
    out = {}
    p1, p2, _, _ = _two_view(n, 0.25, 31, 0.0005)
    out[4] = (p1, p2, 0.002)
    A = rng.uniform(-4, 4, (n, 3))
    B = 1.7 * A @ _rot([1, 2, 3], 0.7).T + np.array([1.0, -2.0, 0.5]) + rng.normal(size=(n, 3)) * 0.002
    bad = rng.random(n) < 0.3
    B[bad] += rng.uniform(0.5, 2, (int(bad.sum()), 3))
    out[5] = (A, B, 0.02)
    nrm = np.array([0.2, -0.3, 0.93]); nrm /= np.linalg.norm(nrm)
    P = rng.uniform(-5, 5, (n, 3))
    P -= np.outer(P @ nrm + 1.5, nrm)
    P += np.outer(rng.normal(size=n) * 0.002, nrm)
    P[bad] += np.outer(rng.uniform(0.2, 2, int(bad.sum())), nrm)
    out[6] = (P, P, 0.01)
    X = np.c_[rng.uniform(-3, 3, (n, 2)), rng.uniform(-1, 1, n)]
    Xc = X @ _rot([0.3, -1, 0.2], 0.4).T + np.array([0.2, -0.1, 6.0])
    uv = Xc[:, :2] / Xc[:, 2:3] + rng.normal(size=(n, 2)) * 0.0003
    uv[bad] += rng.uniform(0.03, 0.2, (int(bad.sum()), 2))
    out[7] = (X, uv, 0.003)
    return out


@pytest.mark.parametrize("model", [4, 5, 6, 7])
def test_ransac_new_models_bit_exact_vs_oracle(ctx, oracle, model):
    """findEssentialMatrix / findSIM3 / findPlane / findPnP hypotheses (GSLAM/core/Estimator.h:118-161): model doubles and
    inlier masks identical to the oracle's, bit for bit -- the solvers use + - * / sqrt only (Jacobi eigen-decomposition
    for Horn's quaternion and the essential projection, full-pivot elimination for the DLT)."""
    from gslam_amd import estimator
    P, Q, thr = _cases()[model]
    for seed in (1, 999):
        em, emask, ecnt = oracle.ransac(model, P, Q, thr, seed=seed)
        gm, gmask, gcnt = estimator.estimate(ctx, model, P, Q, thr, seed=seed)
        assert gcnt == ecnt and np.array_equal(gmask, emask), (model, gcnt, ecnt)
        assert gm.tobytes() == em.tobytes(), (model, gm, em)
        assert ecnt > 400
    m, mask, cnt = estimator.estimate(ctx, model, P[:2], Q[:2], thr)  # fewer points than the minimal sample
    assert cnt == 0 and not m.any()


def test_triangulate_bit_exact_vs_oracle(ctx, oracle):
    from gslam_amd import estimator
    from gslam_amd.ba_synth import _quat_from_R
    from test_ransac_oracle import _rot
    rng = np.random.default_rng(3)
    R, t = _rot([0.1, 1, 0.2], 0.2), np.array([-0.8, 0.02, 0.05])
    pose = np.r_[_quat_from_R(R[None])[0], t]
    X = np.c_[rng.uniform(-2, 2, (300, 2)), rng.uniform(3, 10, 300)]
    d1 = X / X[:, 2:3] + rng.normal(size=X.shape) * 1e-4
    X2 = X @ R.T + t
    d2 = X2 / X2[:, 2:3]
    d2[::17] = d1[::17] @ R.T  # parallel rays
    d1[5::23, 0] -= 3.0        # diverging rays
    got, ok = estimator.triangulate(ctx, pose, d1, d2)
    poses = np.tile(pose, (300, 1))
    got2, ok2 = estimator.triangulate(ctx, poses, d1, d2)
    assert got.tobytes() == got2.tobytes() and np.array_equal(ok, ok2)
    n_ok = 0
    for i in range(300):
        e, eok = oracle.triangulate(pose, d1[i], d2[i])
        assert eok == ok[i] and e.tobytes() == got[i].tobytes(), i
        n_ok += eok
    assert 200 < n_ok < 300


@pytest.mark.parametrize("model,thr,n,noise", [(0, 2.0, 1500, 0.4), (1, 2.0, 500, 0.3), (2, 1.0, 1200, 0.3), (3, 0.05, 700, 0.005)])
def test_ransac_confidence_bit_exact_vs_oracle(ctx, oracle, model, thr, n, noise):
    """gh_ransac_estimate_conf: same winner, mask, model doubles AND the same number of hypotheses examined as the
    oracle's sequential adaptive RANSAC, for every confidence."""
    from gslam_amd import estimator
    P, Q, inl, _ = _corr(model, n, 0.3, 300 + model + n, noise)
    used_all = []
    for conf in (0.5, 0.95, 0.99, 0.9999999, 1.0, 0.0):
        em, emask, ecnt, eused = oracle.ransac_conf(model, P, Q, thr, conf, seed=9)
        gm, gmask, gcnt, gused = estimator.estimate_conf(ctx, model, P, Q, thr, conf, seed=9)
        assert (gcnt, gused) == (ecnt, eused) and np.array_equal(gmask, emask) and gm.tobytes() == em.tobytes()
        used_all.append(gused)
    assert used_all[-1] == used_all[-2] == 2048 and used_all[0] < 2048


@pytest.mark.parametrize("sampling", [1, 2])
def test_lmeds_and_nosample_bit_exact_vs_oracle(ctx, oracle, sampling):
    """gh_ransac_estimate_ex with GSLAM::EstimatorMethod's LMEDS / NOSAMPLE flags (Estimator.h:86-89), all eight models:
    model doubles, inlier masks and counts identical to the oracle's.  LMedS: the exact radix-select median per hypothesis
    on the GPU vs qsort on the CPU; NOSAMPLE: the sums run in index order on both sides."""
    from gslam_amd import estimator
    from test_ransac_oracle import _all_model_cases
    for model, P, Q, thr, inl in _all_model_cases():
        for t in ((0.0, thr) if sampling == 1 else (thr,)):
            em, emask, ecnt, eused = oracle.estimate_ex(model, P, Q, t, sampling, seed=3)
            gm, gmask, gcnt, gused = estimator.estimate_ex(ctx, model, P, Q, t, sampling, seed=3)
            assert (gcnt, gused) == (ecnt, eused) and np.array_equal(gmask, emask), (model, t, gcnt, ecnt)
            assert gm.tobytes() == em.tobytes(), (model, gm, em)
            assert ecnt > 100 or sampling == 2  # (least squares over 28 % outliers may fit nobody within the threshold)
        if sampling == 2:  # ... and on the inliers alone it fits nearly all of them
            em, emask, ecnt, _ = oracle.estimate_ex(model, P[inl], Q[inl], thr, 2)
            gm, gmask, gcnt, _ = estimator.estimate_ex(ctx, model, P[inl], Q[inl], thr, 2)
            assert gcnt == ecnt and np.array_equal(gmask, emask) and gm.tobytes() == em.tobytes(), model
            assert ecnt >= 0.9 * inl.sum()
    # odd / even counts and undefined errors in the median (a homography that sends points to infinity), tiny inputs
    rng = np.random.default_rng(9)
    for n in (9, 10, 257, 1000):
        P, Q, _, _ = _corr(0, n, 0.3, 500 + n, 0.3)
        em, emask, ecnt, _ = oracle.estimate_ex(0, P, Q, 0.5, sampling, seed=11)
        gm, gmask, gcnt, _ = estimator.estimate_ex(ctx, 0, P, Q, 0.5, sampling, seed=11)
        assert gcnt == ecnt and np.array_equal(gmask, emask) and gm.tobytes() == em.tobytes(), n
    m, mask, cnt, used = estimator.estimate_ex(ctx, 0, np.zeros((3, 2)), np.zeros((3, 2)), 1.0, sampling)
    assert cnt == 0 and not m.any()
