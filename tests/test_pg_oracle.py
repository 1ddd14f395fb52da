"""Pose-graph / alignment oracle (oracle/pg_oracle.c): the SIM3 algebra is PINNED to the reference (live through
oracle/_ref where /root/reference exists, and through tests/golden/sim3_reference.npz everywhere); the solvers are
unpinned (interface only upstream, Optimizer.h:127-148,210-225) and cross-checked here against independent
implementations: scipy.linalg.logm on 4 x 4 similarity matrices, scipy.optimize.least_squares, numpy's SVD Umeyama."""
import os

import numpy as np
import pytest

import oracle_lib
from gslam_amd.pg_synth import make_pose_graph, sim3_inv, sim3_mul

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sim3_reference.npz")


def test_sim3_algebra_equals_the_reference_golden(oracle):
    g = np.load(GOLD)
    for i in range(len(g["mu"])):
        assert np.abs(oracle.sim3_exp(g["mu"][i]) - g["sims"][i]).max() < 1e-12
        assert np.abs(oracle.sim3_log(g["sims"][i]) - g["logs"][i]).max() < 2e-12
        assert np.abs(oracle.sim3_mul(g["sims"][i], g["sims"][(i + 1) % 64]) - g["muls"][i]).max() < 1e-12
        # S * p = R (s p) + t (SIM3.h:120-123)
        S = g["sims"][i]
        from gslam_amd.pg_synth import _qrot
        assert np.abs(_qrot(S[:4], S[7] * g["pts"][i]) + S[4:7] - g["app"][i]).max() < 1e-11
    # the reference's own round trip (TransformTest.cpp:72-96) holds for the restatement too, incl. where it is fragile
    for mu in ([1, -2, 3, 0, 0, 0, 0], [1, 2, 3, 1e-9, 0, 0, 1e-9], [0.1, 0.2, 0.3, 0.5, -0.4, 0.3, 1e-7], [5, 5, 5, 3.0, 0.2, 0.1, -2.0]):
        mu = np.array(mu, float)
        assert np.abs(oracle.sim3_log(oracle.sim3_exp(mu)) - mu).max() < 1e-11
    S = oracle.sim3_exp(np.array([1, 2, 3, 0.3, -0.2, 0.5, 0.4]))
    assert np.abs(oracle.sim3_mul(S, oracle.sim3_inv(S)) - np.array([0, 0, 0, 1, 0, 0, 0, 1.0])).max() < 1e-14


@pytest.mark.skipif(not oracle_lib.have_reference(), reason="needs oracle/_ref (built where /root/reference exists)")
def test_sim3_algebra_equals_the_reference_live(oracle):
    ref = oracle_lib.load_reference()
    rng = np.random.default_rng(9)
    for _ in range(200):
        mu = rng.normal(size=7) * np.array([4, 4, 4, 0.8, 0.8, 0.8, 0.6])
        if abs(mu[6]) < 1e-3:
            mu[6] = 0.01  # the reference computes (s - 1) / sigma without expm1 and loses digits for tiny sigma
        S = ref.sim3_exp(mu)
        assert np.abs(oracle.sim3_exp(mu) - S).max() < 1e-12
        assert np.abs(oracle.sim3_log(S) - ref.sim3_log(S)).max() < 1e-11
    T = ref.se3_exp(np.array([0.3, -0.2, 0.5, 0.4, 0.1, -0.7]))  # tx ty tz qx qy qz qw in the SE3 shim
    pose = np.concatenate([T[3:], T[:3]])
    assert np.abs(oracle.se3_log(pose) - ref.se3_log(T)).max() < 1e-13


def _mat(S):
    from scipy.spatial.transform import Rotation
    M = np.eye(4)
    M[:3, :3] = S[7] * Rotation.from_quat(S[:4]).as_matrix()
    M[:3, 3] = S[4:7]
    return M


def test_sim3_log_and_residuals_against_scipy_logm(oracle):
    """log of the 4 x 4 similarity matrix [[s R, t], [0, 1]] is [[sigma I + [w]x, v], [0, 0]]: scipy.linalg.logm shares no
    code with the closed forms."""
    from scipy.linalg import logm
    rng = np.random.default_rng(4)
    for _ in range(30):
        mu = rng.normal(size=7) * np.array([2, 2, 2, 0.6, 0.6, 0.6, 0.4])
        S = oracle.sim3_exp(mu)
        L = np.real(logm(_mat(S)))
        got = np.array([L[0, 3], L[1, 3], L[2, 3], L[2, 1], L[0, 2], L[1, 0], L[0, 0]])
        assert np.abs(got - mu).max() < 1e-9
        Sj = oracle.sim3_exp(rng.normal(size=7) * 0.5)
        Mm = oracle.sim3_exp(rng.normal(size=7) * 0.5)
        r = oracle.pg_edge_residual(1, S, Sj, Mm)
        E = np.linalg.inv(_mat(Mm)) @ np.linalg.inv(_mat(S)) @ _mat(Sj)
        Le = np.real(logm(E))
        assert np.abs(r - np.array([Le[0, 3], Le[1, 3], Le[2, 3], Le[2, 1], Le[0, 2], Le[1, 0], Le[0, 0]])).max() < 1e-9


@pytest.mark.parametrize("kind,gps", [("se3", 0), ("sim3", 0), ("mixed", 5), ("se3", 4)])
def test_pose_graph_noise_free_recovers_the_truth(oracle, kind, gps):
    truth, start, dof, prob = make_pose_graph(24, 5, kind=kind, seed=3, perturb=0.08, scale_drift=0.2, gps_every=gps)
    S, sm, st = oracle.pg_solve(start, dof, prob, oracle_lib.ba_options(max_iterations=60))
    assert st == 0 and sm.final_cost < 1e-16 * max(1.0, sm.initial_cost) + 1e-18
    # q and -q are the same rotation
    sign = np.sign((S[:, :4] * truth[:, :4]).sum(axis=1))[:, None]
    assert np.abs(S[:, :4] * sign - truth[:, :4]).max() < 1e-7 and np.abs(S[:, 4:] - truth[:, 4:]).max() < 1e-6
    assert S[0].tobytes() == start[0].tobytes()  # the gauge frame is untouched


@pytest.mark.parametrize("kind", ["se3", "sim3"])
def test_pose_graph_optimum_matches_scipy_least_squares(oracle, kind):
    """Noisy measurements with information matrices: the oracle's LM and scipy's trust-region solver (its own finite
    differences, residuals through scipy.linalg.logm) reach the same cost and the same keyframes."""
    from scipy.linalg import expm, logm
    from scipy.optimize import least_squares
    truth, start, dof, prob = make_pose_graph(10, 3, kind=kind, seed=5, noise=0.03, perturb=0.05, scale_drift=0.1, with_info=True)
    S, sm, st = oracle.pg_solve(start, dof, prob, oracle_lib.ba_options(max_iterations=100))
    assert st == 0 and sm.final_cost < sm.initial_cost
    key = "sim3" if kind == "sim3" else "se3"
    first, second, meas, info = prob[key]
    dim = 7 if kind == "sim3" else 6
    chol = [np.linalg.cholesky(info[k].reshape(dim, dim)).T for k in range(len(first))]  # r^T L r = |chol r|^2
    M0 = [_mat(s) for s in start]
    Mm = [np.linalg.inv(_mat(np.concatenate([m, [1.0]]) if len(m) == 7 else m)) for m in meas]

    def hat(d):
        G = np.zeros((4, 4))
        G[:3, :3] = np.array([[d[6], -d[5], d[4]], [d[5], d[6], -d[3]], [-d[4], d[3], d[6]]])
        G[:3, 3] = d[:3]
        return G

    def fun(x):
        Ms = [M0[0]] + [M0[i] @ expm(hat(np.concatenate([x[dim * (i - 1): dim * i], [0.0] * (7 - dim)]))) for i in range(1, len(M0))]
        out = []
        for k in range(len(first)):
            A, B = Ms[first[k]].copy(), Ms[second[k]].copy()
            if kind != "sim3":  # SE3 edges see (R, t) only
                for Q in (A, B):
                    Q[:3, :3] /= np.cbrt(np.linalg.det(Q[:3, :3]))
            L = np.real(logm(Mm[k] @ np.linalg.inv(A) @ B))
            r = np.array([L[0, 3], L[1, 3], L[2, 3], L[2, 1], L[0, 2], L[1, 0], L[0, 0]])[:dim]
            out.append(chol[k] @ r)
        return np.concatenate(out)

    res = least_squares(fun, np.zeros(dim * (len(start) - 1)), method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-12)
    assert abs(0.5 * float(res.fun @ res.fun) - sm.final_cost) <= 1e-6 * sm.final_cost
    assert abs(oracle.pg_cost(S, prob) - sm.final_cost) <= 1e-12 * sm.final_cost


def test_alignment_against_numpy_umeyama(oracle):
    rng = np.random.default_rng(8)
    for with_scale in (True, False):
        src = rng.normal(size=(300, 3)) * 3
        S = oracle.sim3_exp(np.array([1.0, -2.0, 0.5, 0.4, -0.3, 0.8, 0.5 if with_scale else 0.0]))
        from gslam_amd.pg_synth import _qrot
        dst = np.stack([_qrot(S[:4], S[7] * p) + S[4:7] for p in src]) + rng.normal(size=(300, 3)) * 0.01
        ok, out, info, ssq = oracle.align_sim3(src, dst, dof=127 if with_scale else 63)
        assert ok
        # Umeyama 1991 (SVD of the cross-covariance), written down here independently
        ma, mb = src.mean(0), dst.mean(0)
        A, B = src - ma, dst - mb
        U, D, Vt = np.linalg.svd(B.T @ A / len(src))
        W = np.diag([1, 1, np.sign(np.linalg.det(U) * np.linalg.det(Vt))])
        R = U @ W @ Vt
        from scipy.spatial.transform import Rotation
        Ro = Rotation.from_quat(out[:4]).as_matrix()
        assert np.abs(Ro - R).max() < 1e-9
        # Horn's symmetric scale sqrt(sum |b|^2 / sum |a|^2) (the specification) vs Umeyama's trace form: equal up to noise
        s_um = (D * np.diag(W)).sum() / A.var(0).sum() if with_scale else 1.0
        assert abs(out[7] - s_um) < 2e-4 and (with_scale or out[7] == 1.0)
        assert np.abs(out[4:7] - (mb - out[7] * Ro @ ma)).max() < 1e-9
        res = dst - (out[7] * src @ Ro.T + out[4:7])
        assert abs(ssq - (res ** 2).sum()) < 1e-9
        # information = J^T J of dst - S exp(delta) src: finite differences through the oracle's own retraction
        J = np.zeros((3 * len(src), 7))
        for k in range(7):
            if not with_scale and k == 6:
                continue
            d = np.zeros(7)
            d[k] = 1e-6
            Sp, Sm = oracle.sim3_retract(out, d), oracle.sim3_retract(out, -d)
            fp = np.stack([_qrot(Sp[:4], Sp[7] * p) + Sp[4:7] for p in src])
            fm = np.stack([_qrot(Sm[:4], Sm[7] * p) + Sm[4:7] for p in src])
            J[:, k] = -((fp - fm) / 2e-6).reshape(-1)
        assert np.abs(info - J.T @ J).max() <= 1e-5 * np.abs(info).max()
    ok, _, _, _ = oracle.align_sim3(np.zeros((5, 3)), np.zeros((5, 3)))
    assert not ok
