"""GPU parity of gh_graph_solve (general BundleGraph: SIM3 keyframes, pose edges, XYZ and inverse-depth landmarks) against
oracle/graph_oracle.c through the C ABI.  f64 on both sides; the GPU's sums are REPRODUCIBLE (pre-rounded accumulation,
gh_ba_options.deterministic = 1, the default) but in another order than the oracle's: identical LM decisions, every cost of
the trace to 1e-9 relative, bit-identical from one GPU run to the next."""
import os

import numpy as np
import pytest

import oracle_lib
from gslam_amd.pg_synth import make_landmark_graph, make_pose_graph
from lm_trace import assert_identical_trace, assert_same_trace

pytestmark = pytest.mark.gpu


def _opts(huber, iters=40):
    from gslam_amd.ba import default_options
    o = default_options()
    o.huber_delta = huber
    o.max_iterations = iters
    return o


def _compare(ctx, oracle, start, dof, problem, huber, iters=40, rtol=1e-9, state_atol=1e-8):
    """The whole LM trace: same length, same decisions, every cost to rtol; and the GPU run twice is bit-identical.
    States at the bars of SURVEY.md 8(c) since round 6 (frames 1e-8, landmarks 1e-7): the largest differences measured over
    this file's cases are 1.8e-10 (frames), 1.1e-10 (landmarks), 1.5e-12 (inverse depths) -- GSLAM_TEST_PRINT_DIFFS=1 prints them."""
    from gslam_amd import posegraph
    oo = oracle_lib.ba_options(huber=huber, max_iterations=iters)
    S0, x0, r0, so, st0 = oracle.graph_solve(start, dof, problem, oo)
    S1, x1, r1, sg, st1 = posegraph.solve_graph(ctx, start, dof, problem, _opts(huber, iters))
    assert st0 == 0 and st1 == 0
    assert_identical_trace(sg, so, rtol)
    if os.environ.get("GSLAM_TEST_PRINT_DIFFS"):
        print("DIFF graph: frames %.3e landmarks %.3e rho %.3e" % (np.abs(S1 - S0).max(), np.abs(x1 - x0).max() if x0.size else 0.0, np.abs(r1 - r0).max() if r0.size else 0.0))
    assert np.allclose(S1, S0, atol=state_atol) and np.allclose(x1, x0, atol=10 * state_atol) and np.allclose(r1, r0, rtol=1e-8, atol=1e-10)
    S2, x2, r2, sg2, st2 = posegraph.solve_graph(ctx, start, dof, problem, _opts(huber, iters))
    assert st2 == 0 and S2.tobytes() == S1.tobytes() and x2.tobytes() == x1.tobytes() and r2.tobytes() == r1.tobytes()
    assert list(sg2.trace_cost[:sg2.trace_len]) == list(sg.trace_cost[:sg.trace_len]), "gh_graph_solve is not reproducible run to run"
    return so, sg


@pytest.mark.parametrize("kind,n_xyz,n_idp,pose_edges,with_info,huber", [
    ("se3", 200, 0, False, False, 0.01),      # plain BA through the general path
    ("se3", 0, 200, False, False, 0.01),      # inverse-depth points only
    ("se3", 150, 150, False, True, 0.02),     # both kinds, 2x2 informations
    ("sim3", 120, 120, True, False, 0.01),    # free keyframe scales + pose-graph edges mixed with observations
    ("se3", 100, 100, True, True, 0.0),       # no robust kernel
])
def test_graph_solve_matches_the_oracle(ctx, oracle, kind, n_xyz, n_idp, pose_edges, with_info, huber):
    truth, start, dof, problem = make_landmark_graph(n_frames=12, n_xyz=n_xyz, n_idp=n_idp, kind=kind, seed=21, noise=2e-3,
                                                     pose_edges=pose_edges, with_info=with_info, outliers=0.05, obs_per_point=5)
    so, sg = _compare(ctx, oracle, start, dof, problem, huber)
    assert so.final_cost < (0.2 if huber > 0 else 0.5) * so.initial_cost and so.iterations >= 3  # (outliers stay in the sum without a kernel)


def test_graph_solve_without_landmarks_is_the_pose_graph_solver(ctx, oracle):
    from gslam_amd import posegraph
    truth, start, dof, problem = make_pose_graph(n_frames=30, n_loops=5, kind="sim3", seed=4, noise=0.01, scale_drift=0.1)
    S1, s1, st1 = posegraph.solve(ctx, start, dof, problem, _opts(0.0, 30))
    S2, _, _, s2, st2 = posegraph.solve_graph(ctx, start, dof, problem, _opts(0.0, 30))
    assert st1 == st2 == 0 and s1.iterations == s2.iterations
    assert np.allclose(np.array(s1.trace_cost[:s1.trace_len]), np.array(s2.trace_cost[:s2.trace_len]), rtol=1e-9)
    assert np.allclose(S1, S2, atol=1e-9)


def test_graph_solve_fixed_landmarks_points_behind_and_host_observations(ctx, oracle):
    truth, start, dof, problem = make_landmark_graph(n_frames=9, n_xyz=60, n_idp=60, kind="se3", seed=8, noise=1e-3, obs_per_point=4)
    xyz, xfree = problem["xyz"]
    host, anchor, rho, ifree = problem["idp"]
    xfree = xfree.copy(); xfree[::3] = 0
    ifree = ifree.copy(); ifree[1::4] = 0
    xyz = xyz.copy()
    # one point behind every camera that sees it (dropped observations), one nearly at infinity
    xyz[1] = [0.0, 0.0, -50.0]
    rho = rho.copy(); rho[2] = 1e-7
    prob = dict(problem, xyz=(xyz, xfree), idp=(host, anchor, rho, ifree))
    so, sg = _compare(ctx, oracle, start, dof, prob, 0.01)
    from gslam_amd import posegraph
    S1, x1, r1, _, _ = posegraph.solve_graph(ctx, start, dof, prob, _opts(0.01))
    assert np.array_equal(x1[::3], xyz[::3]) and np.array_equal(r1[1::4], rho[1::4])
    assert not np.array_equal(x1[2], xyz[2])


def test_graph_solve_larger_window(ctx, oracle):
    """60 keyframes (n = 420: above the single-block regime of the dense solver), 3000 landmarks, 15 000 observations."""
    truth, start, dof, problem = make_landmark_graph(n_frames=60, n_xyz=1500, n_idp=1500, kind="se3", seed=31, noise=1e-3,
                                                     obs_per_point=5, outliers=0.03)
    so, sg = _compare(ctx, oracle, start, dof, problem, 0.01, iters=15)
    assert so.final_cost < 0.2 * so.initial_cost  # (3 % outliers keep their Huber cost)


def test_graph_solve_rejects_bad_arguments(ctx):
    from gslam_amd import hip, posegraph
    truth, start, dof, problem = make_landmark_graph(n_frames=4, n_xyz=5, n_idp=5, seed=1)
    kind, point, frame, xy, info = problem["obs"]
    bad = dict(problem, obs=(kind, point + 100, frame, xy, info))
    with pytest.raises(hip.GslamHipError):
        posegraph.solve_graph(ctx, start, dof, bad)
    host, anchor, rho, free = problem["idp"]
    with pytest.raises(hip.GslamHipError):
        posegraph.solve_graph(ctx, start, dof, dict(problem, idp=(host, anchor, -rho, free)))


@pytest.mark.parametrize("n_xyz,n_idp,pose_edges", [(150, 150, False), (80, 80, True)])
def test_graph_solve_sphere_projection(ctx, oracle, n_xyz, n_idp, pose_edges):
    """PROJECTION_SPHERE: unit-bearing anchors / measurements, tangent-plane residual."""
    truth, start, dof, problem = make_landmark_graph(n_frames=10, n_xyz=n_xyz, n_idp=n_idp, kind="se3", seed=33, noise=2e-3,
                                                     pose_edges=pose_edges, with_info=True, outliers=0.05, obs_per_point=5,
                                                     projection="sphere")
    so, sg = _compare(ctx, oracle, start, dof, problem, 0.01)
    assert so.final_cost < 0.3 * so.initial_cost


def test_graph_solve_atomics_mode_still_agrees(ctx, oracle):
    """deterministic = 0 keeps the plain f64 atomics of rounds 3-4 (the faster assembly): same run up to the settled tail."""
    from gslam_amd import posegraph
    truth, start, dof, problem = make_landmark_graph(n_frames=12, n_xyz=150, n_idp=150, kind="se3", seed=21, noise=2e-3,
                                                     with_info=True, outliers=0.05, obs_per_point=5)
    oo = oracle_lib.ba_options(huber=0.02, max_iterations=40)
    S0, x0, r0, so, st0 = oracle.graph_solve(start, dof, problem, oo)
    o = _opts(0.02, 40)
    o.deterministic = 0
    S1, x1, r1, sg, st1 = posegraph.solve_graph(ctx, start, dof, problem, o)
    assert st0 == 0 and st1 == 0
    assert_same_trace(sg, so, 1e-7)


def test_graph_solve_random_graphs_are_reproducible(ctx):
    """RANDOM small graphs (a fresh draw every run): two GPU solves of the same problem agree bit for bit -- states and every cost
    of the trace (the property the pre-rounded accumulation exists for; no tolerance involved, so no example can be flaky)."""
    from gslam_amd import posegraph
    rng = np.random.default_rng()
    for _ in range(12):
        nf = int(rng.integers(3, 10))
        n_xyz, n_idp = int(rng.integers(0, 40)), int(rng.integers(0, 40))
        if n_xyz + n_idp == 0:
            n_xyz = 5
        sphere = bool(rng.integers(0, 2))
        truth, start, dof, problem = make_landmark_graph(n_frames=nf, n_xyz=n_xyz, n_idp=n_idp, kind="sim3" if rng.integers(0, 2) else "se3",
                                                         seed=int(rng.integers(0, 10 ** 6)), noise=1e-3, pose_edges=bool(rng.integers(0, 2)),
                                                         with_info=bool(rng.integers(0, 2)), obs_per_point=min(4, nf),
                                                         projection="sphere" if sphere else "pinhole")
        huber = float(rng.choice([0.0, 0.01]))
        a = posegraph.solve_graph(ctx, start, dof, problem, _opts(huber, 12))
        b = posegraph.solve_graph(ctx, start, dof, problem, _opts(huber, 12))
        assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes() and a[2].tobytes() == b[2].tobytes()
        assert list(a[3].trace_cost[:a[3].trace_len]) == list(b[3].trace_cost[:b[3].trace_len])
        assert list(a[3].trace_accepted[:a[3].trace_len]) == list(b[3].trace_accepted[:b[3].trace_len])


def test_graph_solve_fuzz_small_graphs(ctx, oracle):
    """Random small graphs (keyframe kind, landmark mix, pose edges, informations, Huber on / off, sphere): the GPU trace equals
    the oracle's -- whole trace, identical decisions, costs to 1e-9."""
    from hypothesis import given, settings, strategies as st

    # derandomize: the oracle sums in another order than the GPU, and on a near-singular toy graph a cost can land within rounding
    # of an LM threshold -- the examples the driver runs are the ones that were run here (the GPU itself is reproducible: the
    # random draw is in test_graph_solve_random_graphs_are_reproducible)
    @settings(max_examples=25, deadline=None, derandomize=True)
    @given(nf=st.integers(3, 9), n_xyz=st.integers(0, 25), n_idp=st.integers(0, 25), sim3=st.booleans(), pose_edges=st.booleans(),
           info=st.booleans(), huber=st.sampled_from([0.0, 0.01]), sphere=st.booleans(), seed=st.integers(0, 10 ** 6))
    def run(nf, n_xyz, n_idp, sim3, pose_edges, info, huber, sphere, seed):
        if n_xyz + n_idp == 0 and (not pose_edges or sphere):  # (a pure pose graph has no projection to speak of)
            n_xyz = 5
        truth, start, dof, problem = make_landmark_graph(n_frames=nf, n_xyz=n_xyz, n_idp=n_idp, kind="sim3" if sim3 else "se3",
                                                         seed=seed, noise=1e-3, pose_edges=pose_edges, with_info=info,
                                                         obs_per_point=min(4, nf), projection="sphere" if sphere else "pinhole")
        _compare(ctx, oracle, start, dof, problem, huber, iters=12)

    run()
