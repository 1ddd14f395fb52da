"""Host-side mirror of the pose-graph / alignment part of the Optimizer plugin for tests and bench.py: gh_pg_solve
(Optimizer::optimize with se3Graph / sim3Graph / gpsGraph edges, GSLAM/core/Optimizer.h:127-148,229) and gh_align_sim3
(optimizeICP / fitSim3, :210-225).  numpy in, numpy out; everything runs in libgslam_hip.so."""
import ctypes as C

import numpy as np

from . import hip
from .ba import default_options


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _fill_pose_part(pr, S, d, problem, keep):
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


def solve(ctx: hip.Context, frames, dof, problem: dict, options=None):
    """problem: {"se3": (first, second, meas n x 7, info n x 36 | None), "sim3": (first, second, meas n x 8, info n x 49 |
    None), "gps": (frame, meas n x 7, info n x 36 | None)} (any subset).  -> (frames n x 8, summary, status)."""
    options = options or default_options()
    S = np.ascontiguousarray(frames, dtype=np.float64).copy()
    d = np.ascontiguousarray(dof, dtype=np.int32)
    keep = [S, d]
    pr = hip.PgProblem()
    _fill_pose_part(pr, S, d, problem, keep)
    sm = hip.BaSummary()
    st = hip.lib.gh_pg_solve(ctx.h, C.byref(pr), C.byref(options), C.byref(sm))
    if st not in (0, 4):
        ctx.check(st)
    return S, sm, st


def solve_graph(ctx: hip.Context, frames, dof, problem: dict, options=None):
    """The general BundleGraph (gh_graph_solve): the pose-edge keys of solve() plus
      "xyz": (points n x 3, free mask | None), "idp": (host, anchor n x 3, rho, free mask | None),
      "obs": (kind, point, frame, xy n x 2, info n x 4 | None)            (gslam_amd.pg_synth.make_landmark_graph).
      "intrinsics": (fx fy cx cy k1 k2 p1 p2 k3, free-parameter bit mask) -- camera self-calibration: `xy` are pixels then
    options.huber_delta = the projection Huber threshold.  -> (frames, xyz, rho, summary, status), or with "intrinsics"
    (frames, xyz, rho, intrinsics, summary, status)."""
    options = options or default_options()
    S = np.ascontiguousarray(frames, dtype=np.float64).copy()
    d = np.ascontiguousarray(dof, dtype=np.int32)
    keep = [S, d]
    gp = hip.GraphProblem()
    _fill_pose_part(gp.pg, S, d, problem, keep)
    xyz, xfree = problem.get("xyz") or (np.zeros((0, 3)), None)
    host, anchor, rho, ifree = problem.get("idp") or (np.zeros(0, np.int32), np.zeros((0, 3)), np.zeros(0), None)
    kind, point, frame, xy, oinfo = problem.get("obs") or (np.zeros(0, np.int32),) * 3 + (np.zeros((0, 2)), None)
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).copy()
    rho = np.ascontiguousarray(rho, dtype=np.float64).copy()
    u8 = lambda v: None if v is None else np.ascontiguousarray(v, dtype=np.uint8)
    i32 = lambda v: np.ascontiguousarray(v, dtype=np.int32)
    f64 = lambda v: None if v is None else np.ascontiguousarray(v, dtype=np.float64)
    xfree, ifree, host, anchor = u8(xfree), u8(ifree), i32(host), f64(anchor)
    kind, point, frame, xy, oinfo = i32(kind), i32(point), i32(frame), f64(xy), f64(oinfo)
    keep += [xyz, rho, xfree, ifree, host, anchor, kind, point, frame, xy, oinfo]
    gp.n_xyz, gp.xyz, gp.xyz_free = len(xyz), _p(xyz), _p(xfree)
    gp.n_idp, gp.idp_host, gp.idp_anchor, gp.idp_rho, gp.idp_free = len(rho), _p(host), _p(anchor), _p(rho), _p(ifree)
    sphere = problem.get("projection") == "sphere"  # then `xy` is n x 3 unit bearings
    gp.n_obs, gp.obs_kind, gp.obs_point, gp.obs_frame, gp.obs_info = len(kind), _p(kind), _p(point), _p(frame), _p(oinfo)
    gp.projection, gp.obs_xy, gp.obs_bearing = (1, None, _p(xy)) if sphere else (0, _p(xy), None)
    cam = None
    if problem.get("intrinsics") is not None:
        cam = np.ascontiguousarray(problem["intrinsics"][0], dtype=np.float64).copy()
        keep.append(cam)
        gp.intrinsics, gp.intrinsics_free = _p(cam), int(problem["intrinsics"][1])
    sm = hip.BaSummary()
    st = hip.lib.gh_graph_solve(ctx.h, C.byref(gp), C.byref(options), C.byref(sm))
    if st not in (0, 4):
        ctx.check(st)
    return (S, xyz, rho, sm, st) if cam is None else (S, xyz, rho, cam, sm, st)


def align_sim3(ctx: hip.Context, src, dst, dof=127):
    """-> (ok, sim3 8, information 7 x 7, sum of squared residuals)."""
    a = np.ascontiguousarray(src, dtype=np.float64)
    b = np.ascontiguousarray(dst, dtype=np.float64)
    out, info, ssq, ok = np.zeros(8), np.zeros(49), C.c_double(), C.c_int()
    ctx.check(hip.lib.gh_align_sim3(ctx.h, _p(a), _p(b), len(a), int(dof), _p(out), _p(info), C.byref(ssq), C.byref(ok)))
    return bool(ok.value), out, info.reshape(7, 7), ssq.value


def bs_symbolic(n_frames, prow, pcol, root_min=128, max_rounds=64):
    """gh_bs_symbolic (host only): the elimination order of the block-sparse pose-graph solver.
    -> dict(pos, ns, nr, round_ptr, colptr, rows, pair_products)."""
    prow = np.ascontiguousarray(prow, dtype=np.int32)
    pcol = np.ascontiguousarray(pcol, dtype=np.int32)
    counts = np.zeros(5, np.int64)
    pos = np.zeros(n_frames, np.int32)
    st = hip.lib.gh_bs_symbolic(n_frames, len(prow), _p(prow), _p(pcol), root_min, max_rounds, _p(pos), _p(counts), None, 0, None,
                                None, 0)
    if st:
        raise RuntimeError("gh_bs_symbolic failed: %d" % st)
    ns, nr, n_rounds, n_slots = (int(v) for v in counts[:4])
    round_ptr = np.zeros(n_rounds + 1, np.int32)
    colptr = np.zeros(ns + 1, np.int32)
    rows = np.zeros(max(n_slots, 1), np.int32)
    st = hip.lib.gh_bs_symbolic(n_frames, len(prow), _p(prow), _p(pcol), root_min, max_rounds, _p(pos), _p(counts), _p(round_ptr),
                                len(round_ptr), _p(colptr), _p(rows), len(rows))
    if st:
        raise RuntimeError("gh_bs_symbolic failed: %d" % st)
    return dict(pos=pos, ns=ns, nr=nr, round_ptr=round_ptr, colptr=colptr, rows=rows[:n_slots], pair_products=int(counts[4]))


def bs_solve(ctx: hip.Context, prow, pcol, diag, off, g, radius=1e30, root_min=128, max_rounds=64):
    """gh_bs_solve_host: (H + clamp(diag) / radius) x = -g with H given by 7 x 7 blocks (column-major).  -> (x, info)."""
    prow = np.ascontiguousarray(prow, dtype=np.int32)
    pcol = np.ascontiguousarray(pcol, dtype=np.int32)
    diag = np.ascontiguousarray(diag, dtype=np.float64)
    off = np.ascontiguousarray(off, dtype=np.float64)
    g = np.ascontiguousarray(g, dtype=np.float64)
    nf = len(diag)
    x = np.zeros(7 * nf)
    info = C.c_int(0)
    ctx.check(hip.lib.gh_bs_solve_host(ctx.h, nf, len(prow), _p(prow), _p(pcol), _p(diag), _p(off), _p(g), float(radius), root_min,
                                       max_rounds, _p(x), C.byref(info)))
    return x, info.value
