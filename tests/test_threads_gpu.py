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
