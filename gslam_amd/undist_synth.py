"""Remap tables in the layout of GSLAM::UndistorterImpl (GSLAM/core/Undistorter.h:120-203) for tests and bench when
the reference's camera code is not available (GPU box): a radial-distortion map evaluated in float64, then the SAME
float32 post-processing the reference applies (integer/fraction split, xx*yy, the four bilinear coefficients)."""
import numpy as np


def make_tables(w_in, h_in, w_out, h_out, k1=-0.28, k2=0.07, zoom=1.05):
    ys, xs = np.mgrid[0:h_out, 0:w_out].astype(np.float64)
    fx, fy, cx, cy = 0.8 * w_in, 0.8 * w_in, w_in / 2.0, h_in / 2.0
    xn = (xs - w_out / 2.0) / (0.8 * w_out) * zoom
    yn = (ys - h_out / 2.0) / (0.8 * w_out) * zoom
    r2 = xn * xn + yn * yn
    d = 1 + k1 * r2 + k2 * r2 * r2
    u, v = fx * xn * d + cx, fy * yn * d + cy
    bad = (u < 0) | (v < 0) | (u >= w_in) | (v >= h_in)
    remapX = np.where(bad, -1, u).astype(np.float32).reshape(-1)
    remapY = np.where(bad, -1, v).astype(np.float32).reshape(-1)
    ui, vi = np.where(bad, 0, u).astype(np.int32).reshape(-1), np.where(bad, 0, v).astype(np.int32).reshape(-1)
    remapFast = np.where(bad.reshape(-1), -1, ui + w_in * vi).astype(np.int32)
    xxi, yyi = remapX.astype(np.int32), remapY.astype(np.int32)  # truncation, as `int xxi = xx`
    xx = (remapX - xxi.astype(np.float32)).astype(np.float32)
    yy = (remapY - yyi.astype(np.float32)).astype(np.float32)
    xxyy = (xx * yy).astype(np.float32)
    one = np.float32(1)
    coef = np.stack([((one - xx) - yy) + xxyy, xx - xxyy, yy - xxyy, xxyy], axis=1).astype(np.float32)
    idx = np.stack([yyi * w_in + xxi, yyi * w_in + xxi + 1, (yyi + 1) * w_in + xxi, (yyi + 1) * w_in + xxi + 1], axis=1)
    b = bad.reshape(-1)
    coef[b] = 0
    idx[b] = 0
    idx = idx.astype(np.int32)  # like the reference's tables, the last row / column may step one past the image
    return {"remapX": remapX, "remapY": remapY, "remapFast": remapFast, "remapIdx": idx, "remapCoef": coef,
            "w_in": w_in, "h_in": h_in, "w_out": w_out, "h_out": h_out}
