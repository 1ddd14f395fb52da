"""Host-side mirror of the Estimator plugin (gslam_amd/plugin/estimator_plugin.cpp): robust model fitting with
inlier masks through gh_ransac_estimate.  Mirrors GSLAM::Estimator::findHomography / findAffine2D / findFundamental /
findAffine3D (GSLAM/core/Estimator.h:100-147)."""
import ctypes as C

import numpy as np

from . import hip

HOMOGRAPHY, AFFINE2D, FUNDAMENTAL, AFFINE3D = 0, 1, 2, 3


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
