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
