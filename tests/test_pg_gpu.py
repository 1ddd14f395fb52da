"""GPU parity of the pose-graph solver and the 3-D alignment (HIP, through the C ABI) vs oracle/pg_oracle.c.
Bars: identical LM iteration count and accept / reject sequence, per-iteration cost 1e-9 relative, keyframes 1e-7
(the Jacobians are central differences with h = 1e-6 through device sin / cos / atan / log / exp, which differ from
glibc's in the last bit: 1e-16 / 2e-6 ~ 5e-11 per Jacobian entry); alignment 1e-11."""
import numpy as np
import pytest

import oracle_lib
from gslam_amd.pg_synth import make_pose_graph

pytestmark = pytest.mark.gpu


def _compare(ctx, oracle, truth, start, dof, prob, max_it=40):
    from gslam_amd import ba, posegraph
    So, so, sto = oracle.pg_solve(start, dof, prob, oracle_lib.ba_options(max_iterations=max_it), threads=4)
    Sg, sg, stg = posegraph.solve(ctx, start, dof, prob, ba.default_options(max_iterations=max_it))
    assert stg == sto == 0
    assert abs(sg.initial_cost - so.initial_cost) <= 1e-12 * max(so.initial_cost, 1e-30)
    assert (sg.iterations, sg.accepted, sg.termination, sg.trace_len) == (so.iterations, so.accepted, so.termination, so.trace_len)
    for i in range(so.trace_len):
        assert sg.trace_accepted[i] == so.trace_accepted[i], i
        assert abs(sg.trace_radius[i] - so.trace_radius[i]) <= 1e-9 * so.trace_radius[i]
        assert abs(sg.trace_cost[i] - so.trace_cost[i]) <= 1e-9 * max(so.trace_cost[i], 1e-30) + 1e-20
    assert np.abs(Sg - So).max() <= 1e-7
    assert Sg[0].tobytes() == start[0].tobytes()
    return Sg, sg


@pytest.mark.parametrize("kind,gps,info", [("se3", 0, False), ("sim3", 0, True), ("mixed", 6, False), ("se3", 5, True)])
def test_pose_graph_parity_noisy(ctx, oracle, kind, gps, info):
    truth, start, dof, prob = make_pose_graph(40, 8, kind=kind, seed=7, noise=0.02, perturb=0.06, scale_drift=0.15,
                                              gps_every=gps, with_info=info)
    Sg, sg = _compare(ctx, oracle, truth, start, dof, prob)
    assert sg.final_cost < 0.5 * sg.initial_cost


def test_pose_graph_noise_free_recovers_truth_and_respects_dof_masks(ctx, oracle):
    from gslam_amd import ba, posegraph
    truth, start, dof, prob = make_pose_graph(30, 6, kind="sim3", seed=2, perturb=0.08, scale_drift=0.2)
    S, sm, st = posegraph.solve(ctx, start, dof, prob, ba.default_options(max_iterations=60))
    sign = np.sign((S[:, :4] * truth[:, :4]).sum(axis=1))[:, None]
    assert st == 0 and np.abs(S[:, :4] * sign - truth[:, :4]).max() < 1e-7 and np.abs(S[:, 4:] - truth[:, 4:]).max() < 1e-6
    # translation-only keyframes keep rotation and scale bit for bit; scale-free keyframes keep their scale
    dof2 = dof.copy()
    dof2[5:10] = 7
    dof2[10:15] = 63
    S2, _, _ = posegraph.solve(ctx, start, dof2, prob, ba.default_options(max_iterations=30))
    assert np.array_equal(S2[5:10, 7], start[5:10, 7]) and np.abs(S2[5:10, :4] - start[5:10, :4]).max() < 1e-15
    assert np.array_equal(S2[10:15, 7], start[10:15, 7]) and np.abs(S2[10:15, :4] - start[10:15, :4]).max() > 1e-4
    _compare(ctx, oracle, truth, start, dof2, prob, max_it=30)


def _ran(ctx, fn, name):
    ctx.prof_enable(True)
    try:
        out = fn()
        names = set(ctx.prof_collect().keys())
    finally:
        ctx.prof_enable(False)
    return out, name in names


@pytest.mark.parametrize("nf,loops,kind,gps,info,root", [(40, 8, "sim3", 0, True, 4), (60, 10, "mixed", 6, False, 8),
                                                         (200, 30, "se3", 25, True, 16)])
def test_pose_graph_block_sparse_parity(ctx, oracle, monkeypatch, nf, loops, kind, gps, info, root):
    """The same bars through the block-sparse solver (rounds of independent keyframes + dense root, bsparse.hip), forced on
    at sizes the dense oracle finishes in seconds."""
    monkeypatch.setenv("GSLAM_HIP_PG_SPARSE_MIN", "0")
    monkeypatch.setenv("GSLAM_HIP_PG_ROOT", str(root))
    truth, start, dof, prob = make_pose_graph(nf, loops, kind=kind, seed=21, noise=0.01, perturb=0.04, scale_drift=0.1,
                                              gps_every=gps, with_info=info)
    _, used = _ran(ctx, lambda: _compare(ctx, oracle, truth, start, dof, prob, max_it=5 if nf > 100 else 40), "bs_factor_cols")
    assert used


@pytest.mark.parametrize("nf,loops,what", [(400, 60, "n = 2800: the single-launch dataflow factorisation"),
                                           (1500, 200, "n = 10 500: the blocked factorisation, right-hand side as the extra row, "
                                                       "segmented back-substitution")])
def test_pose_graph_dense_and_block_sparse_paths_agree(ctx, monkeypatch, nf, loops, what):
    """Both linear solvers under the same LM loop (each is checked against the oracle at sizes the oracle's dense
    factorisation finishes in seconds): identical accept / reject sequence, costs to 1e-9, keyframes to 1e-7; the
    block-sparse path bit for bit the same twice."""
    from gslam_amd import ba, posegraph
    truth, start, dof, prob = make_pose_graph(nf, loops, kind="sim3", seed=5, noise=0.01, perturb=0.03, scale_drift=0.1)
    monkeypatch.setenv("GSLAM_HIP_PG_SPARSE_MIN", "1000000")
    (Sd, sd, std_), used_dense = _ran(ctx, lambda: posegraph.solve(ctx, start, dof, prob, ba.default_options(max_iterations=8)), "pg_damp")
    monkeypatch.setenv("GSLAM_HIP_PG_SPARSE_MIN", "0")
    (Ss, ss, sts), used = _ran(ctx, lambda: posegraph.solve(ctx, start, dof, prob, ba.default_options(max_iterations=8)), "bs_update")
    assert used and used_dense and std_ == sts == 0
    assert (ss.iterations, ss.accepted, ss.trace_len) == (sd.iterations, sd.accepted, sd.trace_len)
    for i in range(sd.trace_len):
        assert ss.trace_accepted[i] == sd.trace_accepted[i]
        assert abs(ss.trace_cost[i] - sd.trace_cost[i]) <= 1e-9 * sd.trace_cost[i] + 1e-20
    assert np.abs(Ss - Sd).max() <= 1e-7
    S2, s2, _ = posegraph.solve(ctx, start, dof, prob, ba.default_options(max_iterations=8))
    assert S2.tobytes() == Ss.tobytes() and list(s2.trace_cost[:s2.trace_len]) == list(ss.trace_cost[:ss.trace_len])
    print("%s: dense %.1f ms, block-sparse %.1f ms per solve call" % (what, sd.total_ms, ss.total_ms))


def test_graph_arena_bit_identical_to_one_allocation_per_array(ctx, monkeypatch):
    """graph_arena.h: the arrays of a solve bump-allocated from the context's grow-only arena and the uploads sent as one
    staged DMA must not change a bit against one hipMalloc / one copy per array (GSLAM_HIP_PG_ARENA=0) -- on the dense and
    the block-sparse path, on an arena that has to grow, one that is reused with the previous solve's contents in it, and
    one that is larger than the solve (a small graph after a large one)."""
    from gslam_amd import ba, posegraph
    graphs = [make_pose_graph(nf, loops, kind=kind, seed=31 + nf, noise=0.01, perturb=0.04, scale_drift=0.1, gps_every=gps,
                              with_info=info)
              for nf, loops, kind, gps, info in ((60, 10, "mixed", 6, True), (500, 70, "sim3", 0, False), (90, 12, "se3", 0, False))]

    def run_all():
        out = []
        for truth, start, dof, prob in graphs:
            S, sm, st = posegraph.solve(ctx, start, dof, prob, ba.default_options(max_iterations=6))
            assert st == 0
            out.append((S.tobytes(), list(sm.trace_cost[:sm.trace_len]), sm.iterations))
        return out

    monkeypatch.setenv("GSLAM_HIP_PG_ARENA", "0")
    ref = run_all()
    monkeypatch.setenv("GSLAM_HIP_PG_ARENA", "1")
    assert run_all() == ref  # arena grows twice, then serves a smaller graph
    assert run_all() == ref  # arena reused as it is


def test_pose_graph_of_6000_keyframes_recovers_the_truth(ctx):
    """A loop-closing sized essential graph (42 000 unknowns; the dense system would be 14 GB): block-sparse by default."""
    from gslam_amd import ba, posegraph
    truth, start, dof, prob = make_pose_graph(6000, 700, kind="sim3", seed=9, perturb=0.02, scale_drift=0.1)
    (S, sm, st), used = _ran(ctx, lambda: posegraph.solve(ctx, start, dof, prob, ba.default_options(max_iterations=30)), "bs_factor_cols")
    assert used and st == 0 and sm.final_cost < 1e-10 * sm.initial_cost
    sign = np.sign((S[:, :4] * truth[:, :4]).sum(axis=1))[:, None]
    assert np.abs(S[:, :4] * sign - truth[:, :4]).max() < 1e-5 and np.abs(S[:, 4:] - truth[:, 4:]).max() < 1e-4
    print("6000 keyframes: %d iterations in %.1f ms (%.2f ms in the linear solves)" % (sm.iterations, sm.total_ms, sm.solve_ms_total))


def test_alignment_parity(ctx, oracle):
    from gslam_amd import posegraph
    from gslam_amd.pg_synth import _qrot
    rng = np.random.default_rng(3)
    for n, dof in ((5000, 127), (300, 63), (3, 127), (1000, 7 | 64)):
        src = rng.normal(size=(n, 3)) * 4
        S = oracle.sim3_exp(np.array([2.0, -1.0, 0.5, 0.3, 0.7, -0.4, 0.3 if dof & 64 else 0.0]))
        dst = np.stack([_qrot(S[:4], S[7] * p) + S[4:7] for p in src]) + rng.normal(size=(n, 3)) * 0.02
        oko, So, io, sso = oracle.align_sim3(src, dst, dof)
        okg, Sg, ig, ssg = posegraph.align_sim3(ctx, src, dst, dof)
        assert okg == oko and np.abs(Sg - So).max() < 1e-11
        assert np.abs(ig - io).max() <= 1e-10 * max(1.0, np.abs(io).max()) and abs(ssg - sso) <= 1e-10 * max(1.0, sso)
    ok, S, _, _ = posegraph.align_sim3(ctx, np.zeros((10, 3)), np.ones((10, 3)))
    assert not ok and S.tolist() == [0, 0, 0, 1, 0, 0, 0, 1]
    ok, _, _, _ = posegraph.align_sim3(ctx, np.zeros((2, 3)), np.ones((2, 3)))
    assert not ok


# ------------------------------------------------------------------------------------------------------------
# through the Optimizer plugin inside a real GSLAM host process (GSLAM's own BundleGraph / SE3Edge / SIM3Edge / GPSEdge)
import os
import struct
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "build", "plugin_host")
LIBDIR = os.path.join(ROOT, "gslam_amd", "lib")


def _host(args):
    if not (os.path.exists(HOST) and os.path.exists(os.path.join(LIBDIR, "libgslam_optimizer.so"))):
        pytest.skip("build/plugin_host or libgslam_optimizer.so missing (run `make plugins` where the GSLAM headers are)")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = LIBDIR + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    return subprocess.run([HOST] + [str(a) for a in args], capture_output=True, text=True, timeout=300, env=env)


@pytest.mark.parametrize("kind,gps,info", [("mixed", 6, True), ("se3", 0, False)])
def test_optimizer_plugin_pose_graph(tmp_path, oracle, kind, gps, info):
    truth, start, dof, prob = make_pose_graph(30, 6, kind=kind, seed=13, noise=0.02, perturb=0.05, scale_drift=0.1,
                                              gps_every=gps, with_info=info)
    inp, out = tmp_path / "pg.bin", tmp_path / "out.bin"
    se3 = prob.get("se3") or (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 7)), None)
    sim3 = prob.get("sim3") or (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 8)), None)
    gp = prob.get("gps") or (np.zeros(0, np.int32), np.zeros((0, 7)), None)
    with open(inp, "wb") as f:
        f.write(np.array([len(start), len(se3[0]), len(sim3[0]), len(gp[0]), 1 if info else 0, 40], np.int32).tobytes())
        f.write(start.astype(np.float64).tobytes() + dof.astype(np.int32).tobytes())
        for first, second, meas, inf, dim in ((se3[0], se3[1], se3[2], se3[3], 6), (sim3[0], sim3[1], sim3[2], sim3[3], 7)):
            f.write(np.asarray(first, np.int32).tobytes() + np.asarray(second, np.int32).tobytes() + np.asarray(meas, np.float64).tobytes())
            if info:
                f.write((np.asarray(inf, np.float64) if inf is not None else np.tile(np.eye(dim).reshape(-1), (len(first), 1))).tobytes())
        f.write(np.asarray(gp[0], np.int32).tobytes() + np.asarray(gp[1], np.float64).tobytes())
        if info:
            f.write((np.asarray(gp[2], np.float64) if gp[2] is not None else np.tile(np.eye(6).reshape(-1), (len(gp[0]), 1))).tobytes())
    r = _host(["pg", LIBDIR, inp, out])
    assert r.returncode == 0 and "pose_graph_optimize=1" in r.stdout, r.stdout + r.stderr
    raw = open(out, "rb").read()
    S = np.frombuffer(raw, np.float64, len(start) * 8, 4).reshape(-1, 8)
    if info:  # edge lists without their own information get the identity in the file: tell the oracle the same
        prob = dict(prob)
        for key, dim in (("se3", 6), ("sim3", 7)):
            if prob.get(key) is not None and prob[key][3] is None:
                prob[key] = prob[key][:3] + (np.tile(np.eye(dim).reshape(-1), (len(prob[key][0]), 1)),)
    So, so, sto = oracle.pg_solve(start, dof, prob, oracle_lib.ba_options(max_iterations=40), threads=4)
    assert sto == 0 and np.abs(S - So).max() <= 1e-7
    assert oracle.pg_cost(S, prob) <= 1.0000001 * so.final_cost


def test_optimizer_plugin_icp_fitsim3_optimizepose(tmp_path, oracle):
    from gslam_amd.pg_synth import _qrot
    rng = np.random.default_rng(21)
    n = 400
    src = rng.normal(size=(n, 3)) * 3
    S = oracle.sim3_exp(np.array([1.0, 2.0, -0.5, 0.2, -0.6, 0.4, 0.25]))
    dst = np.stack([_qrot(S[:4], S[7] * p) + S[4:7] for p in src]) + rng.normal(size=(n, 3)) * 0.01
    # tracking problem: points in frame 1 at depth 1 / rho, camera 2 at T_12 (P_1 = T_12 P_2)
    m = 250
    X1 = np.c_[rng.uniform(-2, 2, (m, 2)), rng.uniform(3, 8, m)]
    T12 = oracle.se3_retract(np.array([0, 0, 0, 1.0, 0, 0, 0]), np.array([0.3, -0.1, 0.2, 0.05, -0.04, 0.03]))
    qc = np.array([-T12[0], -T12[1], -T12[2], T12[3]])
    X2 = np.stack([_qrot(qc, p - T12[4:7]) for p in X1])
    a1 = np.c_[X1[:, :2] / X1[:, 2:3], np.ones(m)]
    a2 = np.c_[X2[:, :2] / X2[:, 2:3], np.ones(m)] * 2.0   # anchors need not be normalised to z = 1
    idp = np.c_[1.0 / X1[:, 2], np.full(m, 0.1)]
    idp[::17, 0] = -1.0                                     # unknown depth: left out
    start = oracle.se3_retract(T12, np.array([0.04, -0.03, 0.02, 0.01, 0.015, -0.01]))
    inp, out = tmp_path / "align.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(np.array([n, 127], np.int32).tobytes() + src.tobytes() + dst.tobytes())
        f.write(np.array([m], np.int32).tobytes() + a1.tobytes() + a2.tobytes() + idp.tobytes() + start.tobytes())
    r = _host(["align", LIBDIR, inp, out])
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(out, "rb").read()
    assert struct.unpack("3i", raw[:12]) == (1, 1, 1)
    v = np.frombuffer(raw, np.float64, offset=12)
    S1, I1, S2, I2, P3, I3 = v[:8], v[8:57].reshape(7, 7), v[57:65], v[65:114].reshape(7, 7), v[114:122], v[122:158].reshape(6, 6)
    ok, So, Io, _ = oracle.align_sim3(src, dst, 127)
    assert ok and np.abs(S1 - So).max() < 1e-11 and np.abs(S2 - So).max() < 1e-11
    assert np.abs(I1 - Io).max() <= 1e-10 * np.abs(Io).max() and np.abs(I2 - Io).max() <= 1e-10 * np.abs(Io).max()
    assert np.abs(S1 - S).max() < 5e-3
    sign = np.sign(P3[:4] @ T12[:4])
    assert np.abs(P3[:4] * sign - T12[:4]).max() < 1e-8 and np.abs(P3[4:7] - T12[4:7]).max() < 1e-7 and P3[7] == 1.0
    keep = idp[:, 0] > 0
    po, _, io, _ = oracle.ba_pnp(X1[keep], a2[keep, :2] / a2[keep, 2:3], start, want_information=True)
    assert np.abs(P3[:7] - po).max() < 1e-8 and np.abs(I3 - io).max() <= 1e-7 * np.abs(io).max()


@pytest.mark.parametrize("kind,pose_edges,info,sphere", [("se3", False, False, False), ("sim3", True, True, False),
                                                         ("se3", False, True, True)])
def test_optimizer_plugin_general_graph_inverse_depth_and_mixed(tmp_path, oracle, kind, pose_edges, info, sphere):
    """Optimizer::optimize on a BundleGraph with invDepths / invDepthObserves, mappoints and (second case) sim3 pose edges
    in the same graph, through the GSLAM plugin (C++ host): keyframes, map points and inverse depths equal the oracle's;
    sigma and fixed vertices come back untouched; anchors / measurements not on the z = 1 plane are normalised."""
    from gslam_amd.pg_synth import make_landmark_graph
    truth, start, dof, prob = make_landmark_graph(n_frames=10, n_xyz=80, n_idp=80, kind=kind, seed=17, noise=2e-3,
                                                  pose_edges=pose_edges, with_info=info, outliers=0.05, obs_per_point=4,
                                                  projection="sphere" if sphere else "pinhole")
    xyz, xfree = prob["xyz"]
    host, anchor, rho, ifree = prob["idp"]
    okind, opoint, oframe, oxy, oinfo = prob["obs"]
    xfree = xfree.copy(); xfree[::7] = 0
    ifree = ifree.copy(); ifree[::5] = 0
    prob = dict(prob, xyz=(xyz, xfree), idp=(host, anchor, rho, ifree))
    huber = 0.01
    inp, out = tmp_path / "graph.bin", tmp_path / "out.bin"
    se3 = prob.get("se3") or (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 7)), None)
    sim3 = prob.get("sim3") or (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 8)), None)
    rng = np.random.default_rng(1)
    with open(inp, "wb") as f:
        f.write(np.array([len(start), len(se3[0]), len(sim3[0]), 0, 0, 40], np.int32).tobytes())
        f.write(start.astype(np.float64).tobytes() + dof.astype(np.int32).tobytes())
        for first, second, meas in ((se3[0], se3[1], se3[2]), (sim3[0], sim3[1], sim3[2])):
            f.write(np.asarray(first, np.int32).tobytes() + np.asarray(second, np.int32).tobytes() + np.asarray(meas, np.float64).tobytes())
        mx, mi = okind == 0, okind == 1
        f.write(np.array([len(xyz), len(rho), int(mx.sum()), int(mi.sum()), 1 if info else 0, 1 if sphere else 0], np.int32).tobytes())
        f.write(np.float64(huber).tobytes())
        f.write(xyz.astype(np.float64).tobytes() + xfree.astype(np.uint8).tobytes())
        zs = rng.uniform(0.5, 2.0, len(rho))  # anchors off the z = 1 plane: the plugin normalises them
        f.write(host.astype(np.int32).tobytes() + (anchor * zs[:, None]).astype(np.float64).tobytes())
        f.write(np.c_[rho, np.full(len(rho), 0.123)].astype(np.float64).tobytes())
        f.write(np.where(ifree != 0, 3, 2).astype(np.int32).tobytes())  # UPDATE_ID_IDEPTHSIGMA / UPDATE_ID_SIGMA only
        for m in (mx, mi):
            zm = rng.uniform(0.5, 2.0, int(m.sum()))
            f.write(opoint[m].astype(np.int32).tobytes() + oframe[m].astype(np.int32).tobytes())
            m3 = oxy[m] if sphere else np.c_[oxy[m], np.ones(int(m.sum()))]  # (scaled: the plugin normalises bearings / anchors)
            f.write((m3 * zm[:, None]).astype(np.float64).tobytes())
            if info:
                f.write(oinfo[m].astype(np.float64).tobytes())
    r = _host(["pg", LIBDIR, inp, out])
    assert r.returncode == 0 and "pose_graph_optimize=1" in r.stdout, r.stdout + r.stderr
    raw = open(out, "rb").read()
    nf, nx, ni = len(start), len(xyz), len(rho)
    S = np.frombuffer(raw, np.float64, nf * 8, 4).reshape(-1, 8)
    X = np.frombuffer(raw, np.float64, nx * 3, 4 + nf * 64).reshape(-1, 3)
    E = np.frombuffer(raw, np.float64, ni * 2, 4 + nf * 64 + nx * 24).reshape(-1, 2)
    # the plugin lists mappoint observations first, then the inverse-depth ones: same order for the oracle
    order = np.concatenate([np.nonzero(mx)[0], np.nonzero(mi)[0]])
    prob_o = dict(prob, obs=(okind[order], opoint[order], oframe[order], oxy[order], None if oinfo is None or not info else oinfo[order]))
    So, xo, ro, so, sto = oracle.graph_solve(start, dof, prob_o, oracle_lib.ba_options(huber=huber, max_iterations=40), threads=4)
    assert sto == 0 and so.final_cost < 0.3 * so.initial_cost
    assert np.abs(S - So).max() <= 1e-6 and np.abs(X - xo).max() <= 1e-5 and np.allclose(E[:, 0], ro, rtol=1e-5)
    assert np.array_equal(E[:, 1], np.full(ni, 0.123))
    assert np.array_equal(X[::7], xyz[::7]) and np.array_equal(E[::5, 0], rho[::5]) and not np.array_equal(E[1::5, 0], rho[1::5])
