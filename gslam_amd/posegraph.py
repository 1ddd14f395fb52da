"""Host-side mirror of the pose-graph / alignment part of the Optimizer plugin for tests and bench.py: gh_pg_solve
(Optimizer::optimize with se3Graph / sim3Graph / gpsGraph edges, GSLAM/core/Optimizer.h:127-148,229) and gh_align_sim3
(optimizeICP / fitSim3, :210-225).  numpy in, numpy out; everything runs in libgslam_hip.so."""
import ctypes as C

import numpy as np

from . import hip
from .ba import default_options


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def solve(ctx: hip.Context, frames, dof, problem: dict, options=None):
    """problem: {"se3": (first, second, meas n x 7, info n x 36 | None), "sim3": (first, second, meas n x 8, info n x 49 |
    None), "gps": (frame, meas n x 7, info n x 36 | None)} (any subset).  -> (frames n x 8, summary, status)."""
    options = options or default_options()
    S = np.ascontiguousarray(frames, dtype=np.float64).copy()
    d = np.ascontiguousarray(dof, dtype=np.int32)
    keep = [S, d]
    pr = hip.PgProblem()
    pr.n_frames, pr.frame_sim3, pr.frame_dof = len(S), _p(S), _p(d)
    for key in ("se3", "sim3"):
        if problem.get(key) is not None:
            f, s, m, inf = problem[key]
            f, s = np.ascontiguousarray(f, dtype=np.int32), np.ascontiguousarray(s, dtype=np.int32)
            m = np.ascontiguousarray(m, dtype=np.float64)
            inf = np.ascontiguousarray(inf, dtype=np.float64) if inf is not None else None
            keep += [f, s, m, inf]
            setattr(pr, "n_" + key, len(f))
            setattr(pr, key + "_first", _p(f)); setattr(pr, key + "_second", _p(s))
            setattr(pr, key + "_meas", _p(m)); setattr(pr, key + "_info", _p(inf))
    if problem.get("gps") is not None:
        f, m, inf = problem["gps"]
        f, m = np.ascontiguousarray(f, dtype=np.int32), np.ascontiguousarray(m, dtype=np.float64)
        inf = np.ascontiguousarray(inf, dtype=np.float64) if inf is not None else None
        keep += [f, m, inf]
        pr.n_gps, pr.gps_frame, pr.gps_meas, pr.gps_info = len(f), _p(f), _p(m), _p(inf)
    sm = hip.BaSummary()
    st = hip.lib.gh_pg_solve(ctx.h, C.byref(pr), C.byref(options), C.byref(sm))
    if st not in (0, 4):
        ctx.check(st)
    return S, sm, st


def align_sim3(ctx: hip.Context, src, dst, dof=127):
    """-> (ok, sim3 8, information 7 x 7, sum of squared residuals)."""
    a = np.ascontiguousarray(src, dtype=np.float64)
    b = np.ascontiguousarray(dst, dtype=np.float64)
    out, info, ssq, ok = np.zeros(8), np.zeros(49), C.c_double(), C.c_int()
    ctx.check(hip.lib.gh_align_sim3(ctx.h, _p(a), _p(b), len(a), int(dof), _p(out), _p(info), C.byref(ssq), C.byref(ok)))
    return bool(ok.value), out, info.reshape(7, 7), ssq.value
