"""Optimizer::magin -- "Convert bundle graph to pose graph" (GSLAM/core/Optimizer.h:230-232; declaration only in the
reference, so the specification is oracle_ba_marginalize in oracle/ba_oracle.c: parity unpinned).

CPU: the oracle's edge information against an independent numpy evaluation (the two-view Hessian assembled from
per-observation Jacobians and reduced with numpy.linalg, instead of the per-point closed form), edge list properties.
GPU: gh_ba_marginalize against the oracle; the pose graph it produces, solved by gh_pg_solve from perturbed poses, comes back
to the poses it was marginalised at.
"""
import ctypes as C

import numpy as np
import pytest

from gslam_amd.ba_synth import make_graph
from test_independent_cpu import _lin


@pytest.fixture(scope="module")
def oracle():
    import oracle_lib
    return oracle_lib.load()


def _small_graph(seed=3, info=False, fixed=False):
    g = make_graph(12, 300, n_obs_per_point=5, seed=seed)
    rng = np.random.default_rng(seed)
    if info:
        M = rng.standard_normal((len(g["obs_cam"]), 2, 2)) * 0.3
        L = M @ M.transpose(0, 2, 1) + np.eye(2)
        g["obs_info"] = np.ascontiguousarray(L.reshape(-1, 4))
    if fixed:
        pf = np.ones(len(g["point_xyz"]), np.uint8)
        pf[rng.choice(len(pf), 40, replace=False)] = 0
        g["point_free"] = pf
    return g


@pytest.mark.parametrize("info,fixed", [(False, False), (True, False), (False, True)])
def test_oracle_information_is_the_two_view_schur_complement(oracle, info, fixed):
    g = _small_graph(info=info, fixed=fixed)
    first, second, shared, lam = oracle.ba_marginalize(g, huber=0.01)
    assert len(first) > 10 and (first < second).all()
    key = first.astype(np.int64) * 12 + second
    assert (np.diff(key) > 0).all()                      # sorted by (first, second), no pair twice
    ocam, opt = np.asarray(g["obs_cam"]), np.asarray(g["obs_point"])
    pf = g.get("point_free")
    worst = 0.0
    for e in range(0, len(first), 3):
        i, j = int(first[e]), int(second[e])
        pts_i = {int(opt[k]): k for k in np.flatnonzero(ocam == i)}
        pts_j = {int(opt[k]): k for k in np.flatnonzero(ocam == j)}
        common = sorted(set(pts_i) & set(pts_j))
        assert len(common) == shared[e]
        free = [p for p in common if pf is None or pf[p]]
        # unknowns: delta_j (6), then 3 per FREE shared point; camera i and the fixed points are constants
        H = np.zeros((6 + 3 * len(free), 6 + 3 * len(free)))
        for p in common:
            for cam, k in ((i, pts_i[p]), (j, pts_j[p])):
                L = g["obs_info"][k].reshape(2, 2) if info else np.eye(2)
                ok, r, w, Jc, Jp, s = _lin(oracle, g["cam_pose"][cam], 63, g["point_xyz"][p], 1, g["obs_xy"][k],
                                           np.ascontiguousarray(g["obs_info"][k]) if info else None, 0.01)
                assert ok == 1
                J = np.zeros((2, H.shape[0]))
                if cam == j:
                    J[:, :6] = Jc
                if p in free:
                    o = 6 + 3 * free.index(p)
                    J[:, o:o + 3] = Jp
                H += J.T @ (w * L) @ J
        ref = H[:6, :6] - (H[:6, 6:] @ np.linalg.solve(H[6:, 6:], H[6:, :6]) if free else 0.0)
        worst = max(worst, np.abs(lam[e] - ref).max() / np.abs(ref).max())
        assert np.allclose(lam[e], lam[e].T, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
        assert np.linalg.eigvalsh((lam[e] + lam[e].T) / 2).min() > -1e-9 * np.abs(ref).max()   # positive semi-definite
    assert worst < 1e-10, worst


def test_oracle_min_shared_filters_edges(oracle):
    g = make_graph(40, 1500, n_obs_per_point=5, seed=2)
    f1, s1, n1, _ = oracle.ba_marginalize(g, min_shared=1)
    th = int(np.median(n1))
    f5, s5, n5, _ = oracle.ba_marginalize(g, min_shared=th)
    assert 0 < len(f5) < len(f1) and (n5 >= th).all() and (n1 >= 1).all()
    keep = n1 >= th
    assert np.array_equal(f1[keep], f5) and np.array_equal(s1[keep], s5)


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def ctx():
    import torch
    from gslam_amd import hip
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["plain", "info", "fixed", "c4"])
def test_marginalize_matches_oracle(ctx, oracle, kind):
    from gslam_amd import ba
    g = make_graph(500, 50000, n_obs_per_point=6, seed=1) if kind == "c4" else _small_graph(info=kind == "info", fixed=kind == "fixed")
    for ms in (1, 8):
        f, s, n, lam = ba.marginalize(ctx, g, huber=0.01, min_shared=ms)
        fo, so, no, lamo = oracle.ba_marginalize(g, huber=0.01, min_shared=ms)
        assert np.array_equal(f, fo) and np.array_equal(s, so) and np.array_equal(n, no)
        scale = np.abs(lamo).reshape(len(lamo), -1).max(axis=1)[:, None, None]
        assert (np.abs(lam - lamo) <= 1e-11 * scale).all()


def _qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def _qrot(q, v):
    qv = np.array([v[0], v[1], v[2], 0.0])
    qc = np.array([-q[0], -q[1], -q[2], q[3]])
    return _qmul(_qmul(q, qv), qc)[:3]


@pytest.mark.gpu
def test_marginalized_pose_graph_recovers_the_poses(ctx, oracle):
    """The edges (T_i^-1 T_j, Lambda_ij) form a pose graph whose minimum is the state they were built at: perturbed
    keyframes solved by gh_pg_solve (first keyframe fixed) come back to it."""
    from gslam_amd import ba, posegraph
    g = make_graph(40, 3000, n_obs_per_point=6, seed=4)
    poses = np.asarray(g["cam_pose"], dtype=np.float64)
    f, s, n, lam = ba.marginalize(ctx, g, huber=0.01, min_shared=10)
    assert len(f) >= 39
    meas = np.zeros((len(f), 7))
    for e, (i, j) in enumerate(zip(f, s)):
        qi, ti, qj, tj = poses[i, :4], poses[i, 4:], poses[j, :4], poses[j, 4:]
        qic = np.array([-qi[0], -qi[1], -qi[2], qi[3]])
        meas[e, :4] = _qmul(qic, qj)
        meas[e, 4:] = _qrot(qic, tj - ti)
    frames = np.concatenate([poses, np.ones((len(poses), 1))], axis=1)
    rng = np.random.default_rng(0)
    start = frames.copy()
    start[1:, 4:7] += 0.05 * rng.standard_normal((len(poses) - 1, 3))
    for k in range(1, len(poses)):
        w = 0.02 * rng.standard_normal(3)
        dq = np.array([w[0] / 2, w[1] / 2, w[2] / 2, 1.0])
        q = _qmul(start[k, :4], dq / np.linalg.norm(dq))
        start[k, :4] = q / np.linalg.norm(q)
    dof = np.full(len(poses), 63, np.int32)
    dof[0] = 0
    out, sm, st = posegraph.solve(ctx, start, dof, {"se3": (f, s, meas, lam.reshape(len(f), 36))},
                                  ba.default_options(max_iterations=50))
    # the cost collapses (every residual vanishes in the metric of its information) ...
    assert sm.initial_cost > 1.0 and sm.final_cost < 1e-9 * sm.initial_cost, (sm.initial_cost, sm.final_cost)
    # ... the well-constrained part of the state is back where it was: rotations to 1e-4 rad; translations only as far as
    # two views constrain them (a two-view information has no stiffness along the baseline's scale)
    sign = np.sign(np.sum(out[:, :4] * frames[:, :4], axis=1))[:, None]
    assert np.abs(out[:, :4] * sign - frames[:, :4]).max() < 1e-4
    assert np.abs(out[:, 4:7] - frames[:, 4:7]).max() < 0.05 * 0.5
