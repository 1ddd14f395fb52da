"""GSLAM runs every plugin in its own thread (GSLAM/gslam/main.cpp:45, Messenger worker pools): two host threads,
each with its own gh_ctx (own stream), must work concurrently without interfering."""
import threading

import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu


def test_two_contexts_two_threads(oracle):
    import ctypes as C
    from gslam_amd import hip
    from gslam_amd.orb import KP_DTYPE
    g = [oracle.synth_frame(640, 480, 900 + i) for i in range(2)]
    exp = [oracle.orb_extract(x, 800) for x in g]
    q = [oracle_lib.random_descriptors(700, 50 + i) for i in range(2)]
    t = [oracle_lib.random_descriptors(900, 60 + i) for i in range(2)]
    exp_m = [oracle.bf_match(q[i], t[i]) for i in range(2)]
    errors = []

    def worker(i):
        try:
            ctx = hip.Context(0)  # private stream
            prm = hip.OrbParams(800, 8, 20, 7)
            plan = C.c_void_p()
            ctx.check(hip.lib.gh_orb_plan_create(ctx.h, 640, 480, 1, C.byref(prm), C.byref(plan)))
            pv = lambda a: a.ctypes.data_as(C.c_void_p)
            for rep in range(20):
                kps = np.zeros(800, KP_DTYPE)
                desc = np.zeros((800, 32), np.uint8)
                n = C.c_int32()
                ctx.check(hip.lib.gh_orb_extract_host(plan, pv(g[i]), 640, pv(kps), pv(desc), C.byref(n)))
                assert n.value == len(exp[i][0]) and kps[:n.value].tobytes() == exp[i][0].tobytes()
                assert np.array_equal(desc[:n.value], exp[i][1])
                idx1, d1, d2 = np.empty(700, np.int32), np.empty(700, np.uint16), np.empty(700, np.uint16)
                ctx.check(hip.lib.gh_bf_match_host(ctx.h, pv(q[i]), 700, pv(t[i]), 900, pv(idx1), pv(d1), pv(d2)))
                assert np.array_equal(idx1, exp_m[i][0]) and np.array_equal(d1, exp_m[i][1])
            hip.lib.gh_orb_plan_destroy(plan)
            ctx.close()
        except Exception as e:  # surfaced in the main thread
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errors, errors


def test_one_context_shared_by_four_threads(oracle):
    """A single gh_ctx (created on the main thread) used concurrently from Messenger-style worker threads: every entry
    point serialises on the context and binds its device (GH_ENTER), so the shared scratch / pinned blocks and the
    BA arena cannot be torn.  ctypes releases the GIL during the calls, so the threads really overlap."""
    import ctypes as C
    from gslam_amd import ba, hip
    from gslam_amd.ba_synth import make_graph
    ctx = hip.Context(0)
    q = [oracle_lib.random_descriptors(500 + 37 * i, 70 + i) for i in range(4)]
    t = [oracle_lib.random_descriptors(800 + 91 * i, 80 + i) for i in range(4)]
    exp_m = [oracle.bf_match(q[i], t[i]) for i in range(4)]
    graphs = [make_graph(6, 80 + 10 * i, n_obs_per_point=4, seed=90 + i) for i in range(4)]
    exp_b = [oracle.ba_solve(g, oracle_lib.ba_options(max_iterations=15)) for g in graphs]
    errors = []
    pv = lambda a: a.ctypes.data_as(C.c_void_p)

    def worker(i):
        try:
            for rep in range(15):
                nq, nt = len(q[i]), len(t[i])
                idx1, d1, d2 = np.empty(nq, np.int32), np.empty(nq, np.uint16), np.empty(nq, np.uint16)
                ctx.check(hip.lib.gh_bf_match_host(ctx.h, pv(q[i]), nq, pv(t[i]), nt, pv(idx1), pv(d1), pv(d2)))
                assert np.array_equal(idx1, exp_m[i][0]) and np.array_equal(d1, exp_m[i][1]) and np.array_equal(d2, exp_m[i][2])
                poses, pts, s, st = ba.solve(ctx, graphs[i], ba.default_options(max_iterations=15))
                assert st == 0 and s.iterations == exp_b[i][2].iterations
                assert np.abs(poses - exp_b[i][0]).max() < 1e-8 and np.abs(pts - exp_b[i][1]).max() < 1e-8
        except Exception as e:
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    [x.start() for x in th]
    [x.join() for x in th]
    ctx.close()
    assert not errors, errors
