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


def test_pose_graph_larger_essential_graph(ctx, oracle):
    """400 keyframes, 460 SIM3 edges: n = 2800 unknowns, the dataflow factorisation path of the dense solver."""
    truth, start, dof, prob = make_pose_graph(400, 60, kind="sim3", seed=11, noise=0.01, perturb=0.03, scale_drift=0.1)
    _compare(ctx, oracle, truth, start, dof, prob, max_it=12)


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
