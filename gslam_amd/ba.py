"""Host-side mirror of the Optimizer plugin (gslam_amd/plugin/optimizer_plugin.cpp) for tests and
bench.py: bundle adjustment through gh_ba_solve.  numpy arrays in, numpy arrays out; the solver runs on
the GPU (no CPU fallback).

Mirrors GSLAM::Optimizer::optimize(BundleGraph&) (GSLAM/core/Optimizer.h:229): keyframes are T_wc as
[qx qy qz qw tx ty tz] with UPDATE_KF_* dof bits, mappoints xyz (+ notFixed flag), observations are
(pointId, frameId, normalised xy on the z = 1 plane, optional 2x2 information).
"""
import ctypes as C

import numpy as np

from . import hip


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def default_options(**kw):
    o = hip.BaOptions()
    hip.lib.gh_ba_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def solve(ctx: hip.Context, graph: dict, options=None):
    """graph: dict with cam_pose (Nc x 7), cam_dof (Nc), point_xyz (Np x 3), obs_cam, obs_point, obs_xy
    (+ optional point_free, obs_info).  Returns (poses, points, summary, status)."""
    options = options or default_options()
    poses = np.ascontiguousarray(graph["cam_pose"], dtype=np.float64).copy()
    pts = np.ascontiguousarray(graph["point_xyz"], dtype=np.float64).copy()
    dof = np.ascontiguousarray(graph["cam_dof"], dtype=np.int32)
    ocam = np.ascontiguousarray(graph["obs_cam"], dtype=np.int32)
    opt = np.ascontiguousarray(graph["obs_point"], dtype=np.int32)
    oxy = np.ascontiguousarray(graph["obs_xy"], dtype=np.float64)
    pfree = graph.get("point_free")
    pfree = np.ascontiguousarray(pfree, dtype=np.uint8) if pfree is not None else None
    info = graph.get("obs_info")
    info = np.ascontiguousarray(info, dtype=np.float64) if info is not None else None
    pr = hip.BaProblem(len(poses), len(pts), len(ocam), _ptr(poses), _ptr(dof), _ptr(pts), _ptr(pfree), _ptr(ocam),
                       _ptr(opt), _ptr(oxy), _ptr(info))
    s = hip.BaSummary()
    st = hip.lib.gh_ba_solve(ctx.h, C.byref(pr), C.byref(options), C.byref(s))
    if st not in (0, 4):
        ctx.check(st)
    return poses, pts, s, st


def camera_order(graph: dict, with_points=False):
    """gh_ba_camera_order (host only, no GPU): the camera order gh_ba_solve would use inside the solver.
    Returns (perm [new position -> caller's camera], border cameras, camera span of the band part, reordered)
    [+ border points when with_points: the choice gh_ba_solve makes between a camera border and a point border]."""
    ocam = np.ascontiguousarray(graph["obs_cam"], dtype=np.int32)
    opt = np.ascontiguousarray(graph["obs_point"], dtype=np.int32)
    nc, npnt = len(graph["cam_dof"]), len(graph["point_xyz"])
    pr = hip.BaProblem(nc, npnt, len(ocam), None, None, None, None, _ptr(ocam), _ptr(opt), None, None)
    perm = np.zeros(nc, np.int32)
    nb, span, re, nbp = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    st = hip.lib.gh_ba_camera_order(C.byref(pr), _ptr(perm), C.byref(nb), C.byref(span), C.byref(re),
                                    C.byref(nbp) if with_points else None)
    if st != 0:
        raise ValueError("gh_ba_camera_order: status %d" % st)
    if with_points:
        return perm, nb.value, span.value, bool(re.value), nbp.value
    return perm, nb.value, span.value, bool(re.value)


class Graph:
    """gh_ba_graph_*: the problem stays in HBM between solves (index lists, pair lists, tables, arrays)."""

    def __init__(self, ctx: hip.Context, graph: dict, options=None):
        self.ctx = ctx
        options = options or default_options()
        a = {k: np.ascontiguousarray(graph[k], dtype=dt) for k, dt in (("cam_pose", np.float64), ("point_xyz", np.float64),
             ("cam_dof", np.int32), ("obs_cam", np.int32), ("obs_point", np.int32), ("obs_xy", np.float64))}
        pfree = graph.get("point_free")
        pfree = np.ascontiguousarray(pfree, dtype=np.uint8) if pfree is not None else None
        info = graph.get("obs_info")
        info = np.ascontiguousarray(info, dtype=np.float64) if info is not None else None
        self.nc, self.np_ = len(a["cam_pose"]), len(a["point_xyz"])
        pr = hip.BaProblem(self.nc, self.np_, len(a["obs_cam"]), _ptr(a["cam_pose"]), _ptr(a["cam_dof"]), _ptr(a["point_xyz"]),
                           _ptr(pfree), _ptr(a["obs_cam"]), _ptr(a["obs_point"]), _ptr(a["obs_xy"]), _ptr(info))
        h = C.c_void_p()
        ctx.check(hip.lib.gh_ba_graph_create(ctx.h, C.byref(pr), C.byref(options), C.byref(h)))
        self.g = h

    def close(self):
        if self.g:
            hip.lib.gh_ba_graph_destroy(self.g)
            self.g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update(self, cam_pose=None, point_xyz=None, obs_xy=None, obs_info=None, cam_dof=None, point_free=None):
        c = lambda a, dt: np.ascontiguousarray(a, dtype=dt) if a is not None else None
        arrs = [c(cam_pose, np.float64), c(point_xyz, np.float64), c(obs_xy, np.float64), c(obs_info, np.float64),
                c(cam_dof, np.int32), c(point_free, np.uint8)]
        self.ctx.check(hip.lib.gh_ba_graph_update(self.g, *[_ptr(a) for a in arrs]))

    def solve(self, options=None):
        options = options or default_options()
        s = hip.BaSummary()
        st = hip.lib.gh_ba_graph_solve(self.g, C.byref(options), C.byref(s))
        if st not in (0, 4):
            self.ctx.check(st)
        return s, st

    def read(self):
        poses, pts = np.zeros((self.nc, 7)), np.zeros((self.np_, 3))
        self.ctx.check(hip.lib.gh_ba_graph_read(self.g, _ptr(poses), _ptr(pts)))
        return poses, pts


def pnp(ctx: hip.Context, points_xyz, obs_xy, pose, dof=63, options=None, want_information=False):
    options = options or default_options()
    X = np.ascontiguousarray(points_xyz, dtype=np.float64)
    m = np.ascontiguousarray(obs_xy, dtype=np.float64)
    p = np.ascontiguousarray(pose, dtype=np.float64).copy()
    info = np.zeros(36) if want_information else None
    s = hip.BaSummary()
    st = hip.lib.gh_ba_pnp(ctx.h, _ptr(X), _ptr(m), len(X), _ptr(p), int(dof), C.byref(options), _ptr(info),
                           C.byref(s))
    if st not in (0, 4):
        ctx.check(st)
    return p, s, (info.reshape(6, 6) if want_information else None)


def potrf_solve(ctx: hip.Context, A, b):
    """Dense SPD solve on the GPU (lower Cholesky in place).  A: n x n (symmetric), b: n.  torch is plumbing."""
    import torch
    n = A.shape[0]
    dA = torch.from_numpy(np.asfortranarray(A, dtype=np.float64).T.copy()).cuda()  # column-major bytes
    db = torch.from_numpy(np.ascontiguousarray(b, dtype=np.float64)).cuda()
    info = C.c_int()
    ctx.check(hip.lib.gh_potrf_solve_dev(ctx.h, C.c_void_p(dA.data_ptr()), n, n, C.c_void_p(db.data_ptr()),
                                         C.byref(info)))
    ctx.sync()
    L = np.tril(dA.cpu().numpy().T)
    return L, db.cpu().numpy(), info.value


def band_solve(ctx: hip.Context, A, b, half_bandwidth):
    """Band SPD solve by block cyclic reduction (gh_band_solve_dev).  A: n x n symmetric with A[r][c] = 0 for
    |r - c| > half_bandwidth; b: n.  Returns (x, info)."""
    import torch
    n = A.shape[0]
    lda = (n + 1 + 15) // 16 * 16
    buf = np.zeros((n, lda))  # row c of `buf` = column c of the column-major device matrix
    buf[:, :n] = np.tril(np.asarray(A, dtype=np.float64)).T
    dA = torch.from_numpy(buf).cuda()
    db = torch.from_numpy(np.ascontiguousarray(b, dtype=np.float64)).cuda()
    info = C.c_int()
    ctx.check(hip.lib.gh_band_solve_dev(ctx.h, C.c_void_p(dA.data_ptr()), n, lda, int(half_bandwidth),
                                        C.c_void_p(db.data_ptr()), C.byref(info)))
    ctx.sync()
    return db.cpu().numpy(), info.value


def arrow_solve(ctx: hip.Context, A, b, n_band, half_bandwidth):
    """Arrowhead SPD solve (gh_arrow_solve_dev): the first n_band unknowns of A (n x n symmetric) form a band of
    half_bandwidth, the others are a dense border; b: n.  Returns (x, info)."""
    import torch
    n = A.shape[0]
    lda = (n + 1 + 15) // 16 * 16
    buf = np.zeros((n, lda))  # row c of `buf` = column c of the column-major device matrix
    buf[:, :n] = np.tril(np.asarray(A, dtype=np.float64)).T
    dA = torch.from_numpy(buf).cuda()
    db = torch.from_numpy(np.ascontiguousarray(b, dtype=np.float64)).cuda()
    info = C.c_int()
    ctx.check(hip.lib.gh_arrow_solve_dev(ctx.h, C.c_void_p(dA.data_ptr()), n, lda, int(n_band), int(half_bandwidth),
                                         C.c_void_p(db.data_ptr()), C.byref(info)))
    ctx.sync()
    return db.cpu().numpy(), info.value


def compact_layout(n_band, half_bandwidth, nbr):
    """gh_cr_compact_layout: (lda, m, brow) of the compact columns, or None when the band does not fit the band solver."""
    lda, m, brow = C.c_int(), C.c_int(), C.c_int()
    if not hip.lib.gh_cr_compact_layout(int(n_band), int(half_bandwidth), int(nbr), C.byref(lda), C.byref(m), C.byref(brow)):
        return None
    return lda.value, m.value, brow.value


def to_compact(A, n_band, lda, m, brow):
    """The lower triangle of the symmetric n x n matrix A in the compact layout (include/gslam_hip.h: gh_cr_compact_layout):
    row c of the result = column c of the device matrix.  Elements outside the band / border must be zero in A."""
    A = np.asarray(A, dtype=np.float64)
    n = A.shape[0]
    buf = np.zeros((n, lda))
    for c in range(n):
        if c < n_band:
            J = c // m
            r1 = min(n_band, (J + 2) * m)
            buf[c, c - J * m:r1 - J * m] = A[c:r1, c]
            assert not A[r1:n_band, c].any(), "a band element more than one superblock below its column"
            buf[c, brow:brow + n - n_band] = A[n_band:, c]
        else:
            buf[c, brow + c - n_band:brow + n - n_band] = A[c:, c]
    return buf


def arrow_solve_compact(ctx: hip.Context, A, b, n_band, half_bandwidth):
    """gh_arrow_solve_compact_dev: arrow_solve on the compact layout (n_band == n: a band).  Returns (x, info)."""
    import torch
    n = A.shape[0]
    lda, m, brow = compact_layout(n_band, half_bandwidth, n - n_band)
    dA = torch.from_numpy(to_compact(A, n_band, lda, m, brow)).cuda()
    db = torch.from_numpy(np.ascontiguousarray(b, dtype=np.float64)).cuda()
    info = C.c_int()
    ctx.check(hip.lib.gh_arrow_solve_compact_dev(ctx.h, C.c_void_p(dA.data_ptr()), n, lda, int(n_band), int(half_bandwidth),
                                                 C.c_void_p(db.data_ptr()), C.byref(info)))
    ctx.sync()
    return db.cpu().numpy(), info.value


def marginalize(ctx: hip.Context, graph: dict, huber=0.01, min_shared=1):
    """gh_ba_marginalize (Optimizer::magin): the SE3 edges of the pose graph a bundle graph marginalises to.
    Returns (first, second, shared, info n x 6 x 6)."""
    poses = np.ascontiguousarray(graph["cam_pose"], dtype=np.float64)
    pts = np.ascontiguousarray(graph["point_xyz"], dtype=np.float64)
    dof = np.ascontiguousarray(graph["cam_dof"], dtype=np.int32)
    ocam = np.ascontiguousarray(graph["obs_cam"], dtype=np.int32)
    opt = np.ascontiguousarray(graph["obs_point"], dtype=np.int32)
    oxy = np.ascontiguousarray(graph["obs_xy"], dtype=np.float64)
    pfree = graph.get("point_free")
    pfree = np.ascontiguousarray(pfree, dtype=np.uint8) if pfree is not None else None
    info = graph.get("obs_info")
    info = np.ascontiguousarray(info, dtype=np.float64) if info is not None else None
    pr = hip.BaProblem(len(poses), len(pts), len(ocam), _ptr(poses), _ptr(dof), _ptr(pts), _ptr(pfree), _ptr(ocam),
                       _ptr(opt), _ptr(oxy), _ptr(info))
    n = C.c_int32()
    ctx.check(hip.lib.gh_ba_marginalize(ctx.h, C.byref(pr), float(huber), int(min_shared), 0, None, None, None, None, C.byref(n)))
    ne = n.value
    first, second, shared = (np.zeros(ne, np.int32) for _ in range(3))
    lam = np.zeros((ne, 6, 6))
    if ne:
        ctx.check(hip.lib.gh_ba_marginalize(ctx.h, C.byref(pr), float(huber), int(min_shared), ne, _ptr(first), _ptr(second),
                                            _ptr(shared), _ptr(lam), C.byref(n)))
    return first, second, shared, lam
