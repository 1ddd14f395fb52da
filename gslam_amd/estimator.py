"""Host-side mirror of the Estimator plugin (gslam_amd/plugin/estimator_plugin.cpp): robust model fitting with
inlier masks through gh_ransac_estimate.  Mirrors GSLAM::Estimator::findHomography / findAffine2D / findFundamental /
findAffine3D (GSLAM/core/Estimator.h:100-147)."""
import ctypes as C

import numpy as np

from . import hip

HOMOGRAPHY, AFFINE2D, FUNDAMENTAL, AFFINE3D, ESSENTIAL, SIM3, PLANE, PNP = 0, 1, 2, 3, 4, 5, 6, 7
RANSAC, LMEDS, NOSAMPLE = 0, 1, 2  # GSLAM::EstimatorMethod sampling flags (Estimator.h:86-89) as gh_ransac_estimate_ex takes them


def estimate_ex(ctx: hip.Context, model, src, dst, threshold, sampling, confidence=1.0, seed=1):
    """gh_ransac_estimate_ex -> (model, mask, inliers, hypotheses_used)."""
    src = np.ascontiguousarray(src, dtype=np.float64)
    dst = np.ascontiguousarray(dst, dtype=np.float64)
    n = src.shape[0]
    m = np.zeros(12)
    mask = np.zeros(max(n, 1), np.uint8)
    cnt, used = C.c_int(), C.c_int()
    pv = lambda a: a.ctypes.data_as(C.c_void_p)
    ctx.check(hip.lib.gh_ransac_estimate_ex(ctx.h, int(model), pv(src), pv(dst), n, C.c_double(threshold), C.c_double(confidence),
                                            C.c_uint64(seed), int(sampling), pv(m), pv(mask), C.byref(cnt), C.byref(used)))
    return m, mask[:n].copy(), cnt.value, used.value


def estimate_conf(ctx: hip.Context, model, src, dst, threshold, confidence, seed=1):
    """gh_ransac_estimate_conf -> (model, mask, inliers, hypotheses_used)."""
    src = np.ascontiguousarray(src, dtype=np.float64)
    dst = np.ascontiguousarray(dst, dtype=np.float64)
    n = src.shape[0]
    m = np.zeros(12)
    mask = np.zeros(max(n, 1), np.uint8)
    cnt, used = C.c_int(), C.c_int()
    pv = lambda a: a.ctypes.data_as(C.c_void_p)
    ctx.check(hip.lib.gh_ransac_estimate_conf(ctx.h, int(model), pv(src), pv(dst), n, C.c_double(threshold),
                                              C.c_double(confidence), C.c_uint64(seed), pv(m), pv(mask), C.byref(cnt),
                                              C.byref(used)))
    return m, mask[:n].copy(), cnt.value, used.value


def estimate(ctx: hip.Context, model, src, dst, threshold, seed=1):
    src = np.ascontiguousarray(src, dtype=np.float64)
    dst = np.ascontiguousarray(dst, dtype=np.float64)
    n = src.shape[0]
    m = np.zeros(12)
    mask = np.zeros(max(n, 1), np.uint8)
    cnt = C.c_int()
    pv = lambda a: a.ctypes.data_as(C.c_void_p)
    ctx.check(hip.lib.gh_ransac_estimate(ctx.h, int(model), pv(src), pv(dst), n, C.c_double(threshold), C.c_uint64(seed),
                                         pv(m), pv(mask), C.byref(cnt)))
    return m, mask[:n].copy(), cnt.value


def triangulate(ctx: hip.Context, ref2cur_pose, ref_dir, cur_dir):
    """Midpoint triangulation (GSLAM::Estimator::trianglate); ref2cur_pose: 7 doubles (one pose for all) or n x 7."""
    T = np.ascontiguousarray(ref2cur_pose, dtype=np.float64)
    d1 = np.ascontiguousarray(ref_dir, dtype=np.float64).reshape(-1, 3)
    d2 = np.ascontiguousarray(cur_dir, dtype=np.float64).reshape(-1, 3)
    n = d1.shape[0]
    out = np.zeros((n, 3))
    ok = np.zeros(max(n, 1), np.uint8)
    pv = lambda a: a.ctypes.data_as(C.c_void_p)
    ctx.check(hip.lib.gh_triangulate(ctx.h, pv(T), 7 if T.ndim == 2 else 0, pv(d1), pv(d2), n, pv(out), pv(ok)))
    return out, ok[:n].astype(bool)
