"""GPU parity at the BASELINE.json sizes the round-1 suite stopped short of:
  C2  all-pairs BF on a 50-frame 1920x1080 subset (1225 frame pairs, K = 2000) vs the oracle, bit-exact, on
      descriptors extracted by the HIP path from frames the oracle extracts identically;
  C4  full size (500 cams / 50 k points / 300 k observations, Huber), seeds 1-3: LM iteration count, accept/reject
      sequence, per-iteration cost 1e-9 vs the oracle; final state 1e-8 (STATE_ATOL_FULL);
  C5/10  1000 cams / 100 k points / 600 k observations (n = 6000 reduced system), same bars;
  PnP motion-only BA with noisy matches + outliers + Huber vs oracle_ba_pnp (Optimizer.h:202-207).
The n = 60000 dense-solve residual (C5 full size) lives in test_ba_gpu.py::test_potrf_solve_large_residual_property."""
import os

import numpy as np
import pytest

import oracle_lib
from gslam_amd.ba_synth import make_graph

pytestmark = pytest.mark.gpu

# Full-size bars = the bars of the small graphs (cost 1e-9, state 1e-8, identical accept / reject sequence).
# Rounds 1-3 had to loosen the state to 1e-5 because their generator left 300 of 500 cameras unobserved;
# on the co-visibility-window graphs of gslam_amd/ba_synth.py the oracle compiled with and without FMA contraction differs
# from itself by 1e-11 in pose (tests/test_ba_oracle.py::test_full_c4_state_sensitivity_to_rounding, CPU).
STATE_ATOL_FULL = 1e-8
RADIUS_RTOL_FULL = 1e-9
THREADS = max(1, min(32, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
COST_RTOL = 1e-9
STATE_ATOL = 1e-8


def _u16(t):
    return t.cpu().numpy().view(np.uint16)


def test_c2_all_pairs_50_frames_1080p(ctx, oracle):
    import torch
    from gslam_amd.matcher import BFMatcher
    from gslam_amd.orb import OrbExtractor, kps_to_numpy, synth_frames
    from gslam_amd.sharding import all_pairs_block
    F, W, H, K = 50, 1920, 1080, 2000
    ex = OrbExtractor(ctx, W, H, max_batch=F, n_features=K)
    frames = synth_frames(ctx, F, W, H, base_seed=0x5EED0000, first_frame=400)
    kps, desc, counts = ex.extract(frames)
    torch.cuda.synchronize()
    host = frames.cpu().numpy()
    ok, od, oc = oracle.orb_extract_batch(host, K, threads=THREADS)
    assert np.array_equal(counts.cpu().numpy(), oc)
    assert kps_to_numpy(kps).tobytes() == ok.tobytes()
    assert np.array_equal(desc.cpu().numpy(), od)
    ai, aj = all_pairs_block(0, 1, F, "cuda")
    assert ai.shape[0] == F * (F - 1) // 2
    idx1, d1, d2 = BFMatcher(ctx).match_pairs(desc, counts, ai, aj)
    torch.cuda.synchronize()
    idx1, d1, d2 = idx1.cpu().numpy(), _u16(d1), _u16(d2)
    ai, aj = ai.cpu().numpy(), aj.cpu().numpy()
    for p in range(len(ai)):
        nq, nt = oc[ai[p]], oc[aj[p]]
        e = oracle.bf_match(od[ai[p], :nq], od[aj[p], :nt], threads=THREADS)
        assert np.array_equal(idx1[p, :nq], e[0]), f"pair {p}"
        assert np.array_equal(d1[p, :nq], e[1]) and np.array_equal(d2[p, :nq], e[2]), f"pair {p}"
        assert (idx1[p, nq:] == -1).all()
    ex.close()


def _compare_ba(oracle, ctx, g, max_it, huber=0.01, state_atol=STATE_ATOL_FULL, radius_rtol=RADIUS_RTOL_FULL):
    from gslam_amd import ba
    eo = oracle.ba_solve(g, oracle_lib.ba_options(huber=huber, max_iterations=max_it), threads=THREADS)
    gp = ba.solve(ctx, g, ba.default_options(huber_delta=huber, max_iterations=max_it, deterministic=1))
    so, sg = eo[2], gp[2]
    assert gp[3] == 0 and eo[3] == 0
    assert abs(sg.initial_cost - so.initial_cost) <= COST_RTOL * so.initial_cost
    assert (sg.iterations, sg.accepted, sg.termination, sg.trace_len) == (so.iterations, so.accepted, so.termination,
                                                                         so.trace_len)
    for i in range(so.trace_len):
        assert sg.trace_accepted[i] == so.trace_accepted[i], f"accept/reject differs at iteration {i}"
        assert abs(sg.trace_radius[i] - so.trace_radius[i]) <= radius_rtol * so.trace_radius[i]
        if np.isinf(so.trace_cost[i]):  # candidate rejected: it moved an observation behind its camera
            assert np.isinf(sg.trace_cost[i])
            continue
        assert abs(sg.trace_cost[i] - so.trace_cost[i]) <= COST_RTOL * so.trace_cost[i], f"cost at iteration {i}"
    assert abs(sg.final_cost - so.final_cost) <= COST_RTOL * so.final_cost
    assert np.abs(gp[0] - eo[0]).max() <= state_atol
    assert np.abs(gp[1] - eo[1]).max() <= state_atol
    # the returned state is self-consistent: the oracle's cost AT the GPU's state is the GPU's reported final cost
    assert abs(oracle.ba_cost(g, gp[0], gp[1], huber=huber) - sg.final_cost) <= 1e-12 * sg.final_cost
    return so


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_c4_full_size_parity(ctx, oracle, seed):
    g = make_graph(500, 50000, n_obs_per_point=6, seed=seed)
    assert len(g["obs_cam"]) == 300000
    so = _compare_ba(oracle, ctx, g, max_it=40)
    assert so.termination == 1 and so.iterations >= 8 and so.final_cost < 0.5 * so.initial_cost


def test_c5_tenth_scale_parity(ctx, oracle):
    g = make_graph(1000, 100000, n_obs_per_point=6, seed=1)
    assert len(g["obs_cam"]) == 600000
    so = _compare_ba(oracle, ctx, g, max_it=12)
    assert so.accepted >= 6 and so.final_cost < 0.5 * so.initial_cost


# ---- loop-closure graphs: the arrowhead solver (band + dense border, cameras renumbered inside the solver) against the ORACLE,
# not only against the GPU's own dense path (VERDICT r5 missing #3 / W2).  The oracle solves the caller's graph in the caller's
# camera order with a dense factorisation; the GPU renumbers the far observers of the closure points to the border, solves, and
# hands the poses back in the caller's order (ba.hip: ba_arrow_order / ArrowProblem).  Same bars as the band graphs: identical
# accept / reject sequence, cost 1e-9, state 1e-8.  (GSLAM/core/Optimizer.h:127-148,229.)
@pytest.mark.parametrize("border", ["cameras", "points", "auto"])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_c4_loop_closure_parity_vs_oracle(ctx, oracle, monkeypatch, seed, border):
    """border: the far CAMERAS of the closure points renumbered last (round 5), the closure POINTS kept out of the Schur complement
    (round 6: 3 unknowns each instead of 6 per far camera), or the solver's own choice (the smaller border: the points here)"""
    if border != "auto":
        monkeypatch.setenv("GSLAM_HIP_BA_POINT_BORDER", "1" if border == "points" else "0")
    g = make_graph(500, 50000, n_obs_per_point=6, seed=seed, loop_closures=20)
    assert len(g["closure_points"]) == 20
    so = _compare_ba(oracle, ctx, g, max_it=40)
    assert ctx.last_ba_solver()[0] == "arrow"
    assert ctx.last_ba_border_points() == (0 if border == "cameras" else 20)
    assert (ctx.last_ba_order()[0] > 0) == (border == "cameras")
    assert so.termination == 1 and so.final_cost < 0.5 * so.initial_cost


@pytest.mark.parametrize("border", ["cameras", "points"])
def test_c5_tenth_loop_closure_parity_vs_oracle(ctx, oracle, monkeypatch, border):
    monkeypatch.setenv("GSLAM_HIP_BA_POINT_BORDER", "1" if border == "points" else "0")
    g = make_graph(1000, 100000, n_obs_per_point=6, seed=1, loop_closures=10)
    so = _compare_ba(oracle, ctx, g, max_it=12)
    assert ctx.last_ba_solver()[0] == "arrow" and ctx.last_ba_border_points() == (10 if border == "points" else 0)
    assert so.accepted >= 6 and so.final_cost < 0.5 * so.initial_cost


@pytest.mark.parametrize("border", ["cameras", "points"])
@pytest.mark.parametrize("dense_border", ["0", "1"])
def test_c4_loop_closure_parity_border_structure_forced(ctx, oracle, monkeypatch, dense_border, border):
    monkeypatch.setenv("GSLAM_HIP_BA_POINT_BORDER", "1" if border == "points" else "0")  # (both kinds of border have their block structure)
    """GSLAM_HIP_BA_ARROW_DENSE_BORDER=0: the border kernels skip the (superblock, strip) blocks the host-side propagation marks
    zero (default only from 4 M border entries up); =1: every block treated as dense.  Both against the oracle."""
    monkeypatch.setenv("GSLAM_HIP_BA_ARROW_DENSE_BORDER", dense_border)
    g = make_graph(500, 50000, n_obs_per_point=6, seed=2, loop_closures=20)
    _compare_ba(oracle, ctx, g, max_it=40)
    assert ctx.last_ba_solver()[0] == "arrow"


def test_loop_closure_resident_graph_parity_vs_oracle(ctx, oracle):
    """The same through gh_ba_graph_create / _update / _solve / _read: poses (and the gauge mask) cross the arrow permutation on
    the way in and out; the second solve starts from re-uploaded values in the CALLER's order."""
    from gslam_amd import ba
    g = make_graph(500, 50000, n_obs_per_point=6, seed=3, loop_closures=20)
    eo = oracle.ba_solve(g, oracle_lib.ba_options(huber=0.01, max_iterations=40), threads=THREADS)
    opts = ba.default_options(huber_delta=0.01, max_iterations=40, deterministic=1)
    G = ba.Graph(ctx, g, opts)
    for attempt in range(2):
        if attempt == 1:
            G.update(cam_pose=g["cam_pose"], point_xyz=g["point_xyz"], cam_dof=g["cam_dof"])
        sg, st = G.solve(opts)
        assert st == 0 and ctx.last_ba_solver()[0] == "arrow"
        from lm_trace import assert_identical_trace
        assert_identical_trace(sg, eo[2], rtol=COST_RTOL)
        poses, pts = G.read()
        assert np.abs(poses - eo[0]).max() <= STATE_ATOL_FULL
        assert np.abs(pts - eo[1]).max() <= STATE_ATOL_FULL
    G.close()


def _pnp_case(oracle, seed, n_pts, noise, outlier_every):
    g = make_graph(3, n_pts, n_obs_per_point=3, seed=seed, noise=0.0, outlier_frac=0.0, perturb=False)
    sel = g["obs_cam"] == 1
    X = g["point_xyz_gt"][g["obs_point"][sel]]
    rng = np.random.default_rng(seed)
    m = g["obs_xy"][sel] + rng.standard_normal((int(sel.sum()), 2)) * noise
    if outlier_every:
        m[::outlier_every] += rng.standard_normal((len(m[::outlier_every]), 2)) * 0.1
    start = oracle.se3_retract(g["cam_pose_gt"][1], np.array([0.05, -0.04, 0.03, 0.01, -0.02, 0.015]))
    return X, m, start


@pytest.mark.parametrize("seed,n_pts,noise,outlier_every,huber,dof",
                         [(41, 200, 0.002, 7, 0.01, 63), (42, 1000, 0.002, 5, 0.01, 63), (43, 60, 0.004, 0, 0.0, 63),
                          (44, 300, 0.002, 9, 0.01, 0b111000), (45, 300, 0.002, 9, 0.01, 0b000111)])
@pytest.mark.parametrize("path", ["one_launch", "general_solver"])
def test_pnp_parity_noisy_huber(ctx, oracle, monkeypatch, path, seed, n_pts, noise, outlier_every, huber, dof):
    """Both implementations behind gh_ba_pnp: the single-launch LM kernel (default) and the general solver on a
    1-camera graph (GSLAM_HIP_PNP_KERNEL=0)."""
    from gslam_amd import ba
    monkeypatch.setenv("GSLAM_HIP_PNP_KERNEL", "1" if path == "one_launch" else "0")
    X, m, start = _pnp_case(oracle, seed, n_pts, noise, outlier_every)
    po, so, io, rc = oracle.ba_pnp(X, m, start, dof=dof, opts=oracle_lib.ba_options(huber=huber, max_iterations=50),
                                   want_information=True)
    pg, sg, ig = ba.pnp(ctx, X, m, start, dof=dof, options=ba.default_options(huber_delta=huber, max_iterations=50),
                        want_information=True)
    assert rc == 0
    assert (sg.iterations, sg.accepted, sg.termination, sg.trace_len) == (so.iterations, so.accepted, so.termination,
                                                                         so.trace_len)
    for i in range(so.trace_len):
        assert sg.trace_accepted[i] == so.trace_accepted[i]
        assert abs(sg.trace_cost[i] - so.trace_cost[i]) <= COST_RTOL * so.trace_cost[i] + 1e-20
    assert abs(sg.final_cost - so.final_cost) <= COST_RTOL * so.final_cost
    assert np.abs(pg - po).max() <= STATE_ATOL
    assert np.abs(ig - io).max() <= 1e-9 * np.abs(io).max()
    assert so.final_cost < so.initial_cost
    if dof == 0b111000:  # rotation only: translation bitwise untouched
        assert np.array_equal(pg[4:], start[4:])
    used = "ba_pnp_lm" in _kernels_of(ctx, lambda: ba.pnp(ctx, X, m, start, dof=dof, options=ba.default_options(huber_delta=huber, max_iterations=3)))
    assert used == (path == "one_launch")


def _kernels_of(ctx, fn):
    ctx.prof_enable(True)
    try:
        fn()
        return set(ctx.prof_collect().keys())
    finally:
        ctx.prof_enable(False)


def test_pnp_degenerate_inputs(ctx, oracle):
    """No observation, every point behind the camera, and fewer points than unknowns: same answers as the oracle."""
    from gslam_amd import ba
    start = np.array([0, 0, 0, 1, 0, 0, 0.0])
    for X, m in ((np.zeros((0, 3)), np.zeros((0, 2))), (np.array([[0.1, 0.2, -3.0], [0.0, 0.1, -2.0]]), np.zeros((2, 2))),
                 (np.array([[0.1, 0.2, 3.0], [-0.3, 0.1, 2.0]]), np.array([[0.05, 0.06], [-0.14, 0.06]]))):
        po, so, io, rc = oracle.ba_pnp(X, m, start, dof=63, opts=oracle_lib.ba_options(huber=0.01, max_iterations=20), want_information=True)
        pg, sg, ig = ba.pnp(ctx, X, m, start, dof=63, options=ba.default_options(huber_delta=0.01, max_iterations=20), want_information=True)
        assert (sg.iterations, sg.accepted, sg.termination, sg.trace_len) == (so.iterations, so.accepted, so.termination, so.trace_len)
        assert np.abs(pg - po).max() <= 1e-8 and np.abs(ig - io).max() <= 1e-9 * max(np.abs(io).max(), 1e-300)
        assert abs(sg.final_cost - so.final_cost) <= 1e-9 * so.final_cost + 1e-20
