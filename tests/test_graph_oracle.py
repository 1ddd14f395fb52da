"""The general-BundleGraph oracle (oracle/graph_oracle.c: SIM3 keyframes + pose edges + XYZ and inverse-depth landmarks)
against independent implementations: finite differences of its own residual, scipy.optimize.least_squares on the same
residuals, ba_oracle.c on graphs both can express, pg_oracle.c on pure pose graphs.  The reference holds no
implementation of Optimizer::optimize (GSLAM/core/Optimizer.h:229): parity of this path is UNPINNED, these are the
cross-checks that stand in."""
import numpy as np
import pytest

import oracle_lib
from gslam_amd.pg_synth import make_landmark_graph, make_pose_graph, sim3_mul, _quat_from_rotvec, _qrot


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.Oracle()


def _retract(o, S, d):
    return o.sim3_retract(S, d)


def _residual_py(Sj, Sh, kind, lm, anchor, m):
    """Independent restatement in numpy (world point first, then the camera of frame j)."""
    if kind == 0:
        Xw = np.asarray(lm, float)
    else:
        Xw = Sh[7] * _qrot(Sh[:4], np.asarray(anchor) / lm[0]) + Sh[4:7]
    qc = np.array([-Sj[0], -Sj[1], -Sj[2], Sj[3]])
    Xc = _qrot(qc, Xw - Sj[4:7]) / Sj[7]
    return np.array([Xc[0] / Xc[2] - m[0], Xc[1] / Xc[2] - m[1]]), Xc[2]


def test_observation_residual_and_analytic_jacobians(oracle):
    rng = np.random.default_rng(3)
    for trial in range(40):
        kind = trial & 1
        mk = lambda: np.concatenate([_quat_from_rotvec(rng.normal(size=3) * 0.4), rng.normal(size=3), [np.exp(rng.normal() * 0.3)]])
        Sj, Sh = mk(), mk()
        dof_j, dof_h = (127, 127) if trial % 5 else (0b1011011, 0b0110110)
        if kind == 0:
            Xc = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(2, 6)])
            lm = Sj[7] * _qrot(Sj[:4], Xc) + Sj[4:7]
            anchor = None
        else:
            a = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), 1.0])
            # move frame j so that the point is in front of it
            lm = np.array([rng.uniform(0.1, 0.5)])
            Xw = Sh[7] * _qrot(Sh[:4], a / lm[0]) + Sh[4:7]
            Sj[4:7] = Xw - Sj[7] * _qrot(Sj[:4], np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(2, 6)]))
            anchor = a
        m = rng.normal(size=2) * 0.3
        info = np.array([2.0, 0.3, 0.3, 1.5]) if trial % 3 == 0 else None
        ok, r, w, s, Jj, Jh, Jp = oracle.graph_obs(kind, Sj, dof_j, Sh, dof_h, False, lm, True, anchor, m, info, 0.1)
        assert ok
        r_py, depth = _residual_py(Sj, Sh, kind, lm, anchor, m)
        assert depth > 0 and np.allclose(r, r_py, rtol=1e-11, atol=1e-12)
        L = np.eye(2) if info is None else info.reshape(2, 2)
        assert np.isclose(s, r @ L @ r, rtol=1e-12) and np.isclose(w, 1.0 if s <= 0.01 else 0.1 / np.sqrt(s), rtol=1e-12)
        # central differences along the right-multiplicative deltas and the landmark parameters
        h = 1e-6
        for which, J, dof in ((0, Jj, dof_j), (1, Jh, dof_h)):
            if kind == 0 and which == 1:
                assert not J.any()
                continue
            for k in range(7):
                d = np.zeros(7); d[k] = h
                args_p = (_retract(oracle, Sj, d), Sh) if which == 0 else (Sj, _retract(oracle, Sh, d))
                args_m = (_retract(oracle, Sj, -d), Sh) if which == 0 else (Sj, _retract(oracle, Sh, -d))
                fd = (_residual_py(*args_p, kind, lm, anchor, m)[0] - _residual_py(*args_m, kind, lm, anchor, m)[0]) / (2 * h)
                want = fd if (dof >> k) & 1 else 0.0
                assert np.allclose(J[:, k], want, rtol=2e-6, atol=2e-8), (trial, which, k, J[:, k], fd)
        for k in range(3 if kind == 0 else 1):
            d = np.zeros(len(lm)); d[k] = h
            fd = (_residual_py(Sj, Sh, kind, lm + d, anchor, m)[0] - _residual_py(Sj, Sh, kind, lm - d, anchor, m)[0]) / (2 * h)
            assert np.allclose(Jp[:, k], fd, rtol=2e-6, atol=2e-8)
    # the host observing its own inverse-depth point: constant residual, no Jacobian
    S = np.concatenate([_quat_from_rotvec(np.array([0.1, 0.2, 0.3])), [1.0, 2.0, 3.0], [1.3]])
    ok, r, w, s, Jj, Jh, Jp = oracle.graph_obs(1, S, 127, S, 127, True, np.array([0.2]), True, np.array([0.3, -0.2, 1.0]),
                                               np.array([0.25, -0.1]))
    assert ok and np.allclose(r, [0.05, -0.1]) and not Jj.any() and not Jh.any() and not Jp.any()
    # behind the camera: dropped
    ok = oracle.graph_obs(0, S, 127, S, 127, False, S[4:7] - S[7] * _qrot(S[:4], np.array([0, 0, 1.0])), True, None, np.zeros(2))[0]
    assert not ok


def _scipy_minimum(oracle, start, dof, problem):
    """Minimise the same cost (no robust kernel) with scipy, parametrised by deltas applied to the START state."""
    from scipy.optimize import least_squares
    nf = len(start)
    xyz0, xfree = problem["xyz"]
    host, anchor, rho0, ifree = problem["idp"]
    kind, point, frame, xy, info = problem["obs"]
    cols = [(f, k) for f in range(nf) for k in range(7) if (dof[f] >> k) & 1]

    def unpack(x):
        d = np.zeros((nf, 7))
        for c, (f, k) in enumerate(cols):
            d[f, k] = x[c]
        S = np.stack([oracle.sim3_retract(start[f], d[f]) if d[f].any() else start[f] for f in range(nf)])
        o = len(cols)
        return S, xyz0 + x[o:o + xyz0.size].reshape(-1, 3), rho0 + x[o + xyz0.size:]

    def res(x):
        S, xyz, rho = unpack(x)
        out = []
        for k in range(len(kind)):
            j = frame[k]
            if kind[k] == 0:
                r, _ = _residual_py(S[j], S[j], 0, xyz[point[k]], None, xy[k])
            else:
                p = point[k]
                r, _ = _residual_py(S[j], S[host[p]], 1, rho[p:p + 1], anchor[p], xy[k])
            if info is not None:
                r = np.linalg.cholesky(info[k].reshape(2, 2)).T @ r
            out.append(r)
        for key, t in (("se3", 0), ("sim3", 1)):
            if problem.get(key) is not None:
                f, s, m, inf = problem[key]
                for e in range(len(f)):
                    r = oracle.pg_edge_residual(t, S[f[e]], S[s[e]], m[e])
                    out.append(r if inf is None else np.linalg.cholesky(inf[e].reshape(len(r), len(r))).T @ r)
        return np.concatenate(out)

    x0 = np.zeros(len(cols) + xyz0.size + rho0.size)
    sol = least_squares(res, x0, method="trf", xtol=1e-15, ftol=1e-15, gtol=1e-14, max_nfev=400)
    return 0.5 * float(sol.fun @ sol.fun), unpack(sol.x)


@pytest.mark.parametrize("kind,pose_edges,with_info", [("se3", False, False), ("sim3", True, True), ("se3", True, False)])
def test_graph_solve_reaches_scipys_minimum(oracle, kind, pose_edges, with_info):
    truth, start, dof, problem = make_landmark_graph(n_frames=6, n_xyz=14, n_idp=14, kind=kind, seed=5, noise=2e-3,
                                                     pose_edges=pose_edges, with_info=with_info)
    opts = oracle_lib.ba_options(huber=0.0, max_iterations=80)
    opts.function_tolerance = 1e-14
    S, xyz, rho, sm, st = oracle.graph_solve(start, dof, problem, opts)
    assert st == 0 and sm.final_cost < sm.initial_cost * 0.05
    assert np.isclose(oracle.graph_cost(S, dof, dict(problem, xyz=(xyz, problem["xyz"][1]),
                                                     idp=(problem["idp"][0], problem["idp"][1], rho, problem["idp"][3])), 0.0),
                      sm.final_cost, rtol=1e-12)
    ref_cost, (S2, xyz2, rho2) = _scipy_minimum(oracle, start, dof, problem)
    assert np.isclose(sm.final_cost, ref_cost, rtol=2e-5, atol=1e-12), (sm.final_cost, ref_cost)
    if not (kind == "sim3"):  # (free scales of keyframes that only XYZ points see are a gauge direction)
        assert np.allclose(xyz, xyz2, atol=2e-3) and np.allclose(rho, rho2, atol=2e-4)


def test_graph_solve_equals_ba_oracle_on_a_pure_xyz_graph(oracle):
    """Same residual, same analytic Jacobians (scale 1), same trust-region policy: the traces must agree."""
    truth, start, dof, problem = make_landmark_graph(n_frames=7, n_xyz=60, n_idp=0, kind="se3", seed=9, noise=3e-3, outliers=0.1)
    kind, point, frame, xy, info = problem["obs"]
    g = {"cam_pose": start[:, :7].copy(), "cam_dof": dof & 63, "point_xyz": problem["xyz"][0].copy(), "obs_cam": frame, "obs_point": point,
         "obs_xy": xy, "obs_info": None, "point_free": None}
    opts = oracle_lib.ba_options(huber=0.01, max_iterations=30)
    poses, pts, so = oracle.ba_solve(g, opts)[:3]
    S, xyz, rho, sm, st = oracle.graph_solve(start, dof, problem, opts)
    assert sm.iterations == so.iterations and sm.trace_len == so.trace_len
    assert np.allclose(np.array(sm.trace_cost[:sm.trace_len]), np.array(so.trace_cost[:so.trace_len]), rtol=1e-8)
    assert list(sm.trace_accepted[:sm.trace_len]) == list(so.trace_accepted[:so.trace_len])
    assert np.allclose(S[:, :7], poses, atol=1e-8) and np.allclose(xyz, pts, atol=1e-7)


def test_graph_solve_equals_pg_oracle_on_a_pure_pose_graph(oracle):
    truth, start, dof, problem = make_pose_graph(n_frames=12, n_loops=3, kind="sim3", seed=4, noise=0.01, scale_drift=0.1)
    opts = oracle_lib.ba_options(huber=0.0, max_iterations=30)
    S1, s1, st1 = oracle.pg_solve(start, dof, problem, opts)
    S2, _, _, s2, st2 = oracle.graph_solve(start, dof, problem, opts)
    assert st1 == st2 == 0 and s1.iterations == s2.iterations
    assert np.allclose(np.array(s1.trace_cost[:s1.trace_len]), np.array(s2.trace_cost[:s2.trace_len]), rtol=1e-9)
    assert np.allclose(S1, S2, atol=1e-9)


def test_inverse_depth_recovers_the_truth_and_respects_fixed_landmarks(oracle):
    truth, start, dof, problem = make_landmark_graph(n_frames=8, n_xyz=0, n_idp=80, kind="se3", seed=2, noise=0.0, obs_per_point=5)
    opts = oracle_lib.ba_options(huber=0.0, max_iterations=60)
    opts.function_tolerance = 1e-16
    S, xyz, rho, sm, st = oracle.graph_solve(start, dof, problem, opts)
    assert st == 0 and sm.final_cost < 1e-16
    # the gauge is fixed by frame 0 and one translation component of frame 1: up to that, the truth
    assert np.allclose(rho / problem["truth_rho"], (rho / problem["truth_rho"])[0], rtol=1e-5)
    host, anchor, rho0, free = problem["idp"]
    free = free.copy(); free[:10] = 0
    S, xyz, rho2, sm, st = oracle.graph_solve(start, dof, dict(problem, idp=(host, anchor, rho0, free)), opts)
    assert np.array_equal(rho2[:10], rho0[:10]) and not np.array_equal(rho2[10:], rho0[10:])


def _bearing_residual_py(Sj, Sh, kind, lm, anchor, b):
    """Independent restatement of the sphere residual: tangent-plane coordinates of the predicted bearing."""
    if kind == 0:
        Xw = np.asarray(lm, float)
    else:
        Xw = Sh[7] * _qrot(Sh[:4], np.asarray(anchor) / lm[0]) + Sh[4:7]
    qc = np.array([-Sj[0], -Sj[1], -Sj[2], Sj[3]])
    Xc = _qrot(qc, Xw - Sj[4:7]) / Sj[7]
    y = Xc / np.linalg.norm(Xc)
    k = np.zeros(3); k[int(np.argmin(np.abs(b)))] = 1.0
    e1 = np.cross(b, k); e1 /= np.linalg.norm(e1)
    e2 = np.cross(b, e1)
    return np.array([e1 @ y, e2 @ y]), float(y @ b)


def test_sphere_projection_residual_and_jacobians(oracle):
    rng = np.random.default_rng(11)
    for trial in range(30):
        kind = trial & 1
        mk = lambda: np.concatenate([_quat_from_rotvec(rng.normal(size=3) * 0.5), rng.normal(size=3), [np.exp(rng.normal() * 0.3)]])
        Sj, Sh = mk(), mk()
        Xc = rng.normal(size=3) * 3  # any direction: a panoramic camera sees behind itself too
        if kind == 0:
            lm, anchor = Sj[7] * _qrot(Sj[:4], Xc) + Sj[4:7], None
        else:
            a = rng.normal(size=3); a /= np.linalg.norm(a)
            lm = np.array([rng.uniform(0.1, 0.5)])
            Xw = Sh[7] * _qrot(Sh[:4], a / lm[0]) + Sh[4:7]
            Sj[4:7] = Xw - Sj[7] * _qrot(Sj[:4], Xc)
            anchor = a
        b = Xc / np.linalg.norm(Xc) + rng.normal(size=3) * 0.05
        b /= np.linalg.norm(b)
        ok, r, w, s, Jj, Jh, Jp = oracle.graph_obs(kind, Sj, 127, Sh, 127, False, lm, True, anchor, b, None, 0.0, projection=1)
        r_py, dot = _bearing_residual_py(Sj, Sh, kind, lm, anchor, b)
        assert ok and dot > 0 and np.allclose(r, r_py, rtol=1e-10, atol=1e-12)
        h = 1e-6
        for which, J in ((0, Jj), (1, Jh)):
            if kind == 0 and which == 1:
                continue
            for k in range(7):
                d = np.zeros(7); d[k] = h
                ap = (_retract(oracle, Sj, d), Sh) if which == 0 else (Sj, _retract(oracle, Sh, d))
                am = (_retract(oracle, Sj, -d), Sh) if which == 0 else (Sj, _retract(oracle, Sh, -d))
                fd = (_bearing_residual_py(*ap, kind, lm, anchor, b)[0] - _bearing_residual_py(*am, kind, lm, anchor, b)[0]) / (2 * h)
                assert np.allclose(J[:, k], fd, rtol=2e-6, atol=2e-8), (trial, which, k)
        for k in range(3 if kind == 0 else 1):
            d = np.zeros(len(lm)); d[k] = h
            fd = (_bearing_residual_py(Sj, Sh, kind, lm + d, anchor, b)[0] - _bearing_residual_py(Sj, Sh, kind, lm - d, anchor, b)[0]) / (2 * h)
            assert np.allclose(Jp[:, k], fd, rtol=2e-6, atol=2e-8)
    # opposite hemisphere: dropped
    S = np.concatenate([_quat_from_rotvec(np.array([0.1, 0.2, 0.3])), [1.0, 2.0, 3.0], [1.0]])
    X = S[4:7] + _qrot(S[:4], np.array([0.0, 0.0, 2.0]))
    assert not oracle.graph_obs(0, S, 127, S, 127, False, X, True, None, np.array([0.0, 0.0, -1.0]), None, 0.0, projection=1)[0]


def test_sphere_graph_converges_to_the_truth(oracle):
    truth, start, dof, problem = make_landmark_graph(n_frames=8, n_xyz=40, n_idp=40, kind="se3", seed=6, noise=0.0, obs_per_point=5,
                                                     projection="sphere")
    opts = oracle_lib.ba_options(huber=0.0, max_iterations=60)
    opts.function_tolerance = 1e-16
    S, xyz, rho, sm, st = oracle.graph_solve(start, dof, problem, opts)
    assert st == 0 and sm.final_cost < 1e-15 and sm.initial_cost > 1e-4
    ratio = rho / problem["truth_rho"]
    assert np.allclose(ratio, ratio[0], rtol=1e-5)  # the truth up to the gauge the fixed dof leave
