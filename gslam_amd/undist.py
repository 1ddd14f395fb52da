"""Host-side mirror of GSLAM::Undistorter::undistort / undistortFast (GSLAM/core/Undistorter.h:206-348) on the GPU.
The remap tables are the ones UndistorterImpl::prepareReMap builds on the host."""
import ctypes as C

import numpy as np
import torch

from . import hip


class Undistorter:
    def __init__(self, ctx: hip.Context, tables: dict):
        self.ctx = ctx
        self.w_in, self.h_in, self.w_out, self.h_out = (tables[k] for k in ("w_in", "h_in", "w_out", "h_out"))
        a = lambda k, dt: np.ascontiguousarray(tables[k], dtype=dt)
        x, f, i, c = a("remapX", np.float32), a("remapFast", np.int32), a("remapIdx", np.int32), a("remapCoef", np.float32)
        h = C.c_void_p()
        pv = lambda z: z.ctypes.data_as(C.c_void_p)
        ctx.check(hip.lib.gh_undist_plan_create(ctx.h, self.w_in, self.h_in, self.w_out, self.h_out, pv(x), pv(f), pv(i),
                                                pv(c), C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            hip.lib.gh_undist_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def undistort(self, imgs: torch.Tensor, fast=False, out=None):
        """imgs: B x H_in x W_in [x C] u8 (cuda, dense) -> B x H_out x W_out [x C]."""
        assert imgs.is_cuda and imgs.dtype == torch.uint8 and imgs.is_contiguous()
        B = imgs.shape[0]
        ch = 1 if imgs.dim() == 3 else imgs.shape[3]
        shape = (B, self.h_out, self.w_out) if ch == 1 else (B, self.h_out, self.w_out, ch)
        out = out if out is not None else torch.empty(shape, dtype=torch.uint8, device=imgs.device)
        self.ctx.check(hip.lib.gh_undistort_dev(self.h, C.c_void_p(imgs.data_ptr()), ch, B,
                                                C.c_size_t(self.h_in * self.w_in * ch), C.c_void_p(out.data_ptr()),
                                                C.c_size_t(self.h_out * self.w_out * ch), 1 if fast else 0))
        return out

    def undistort_host(self, img: np.ndarray, fast=False):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        ch = 1 if img.ndim == 2 else img.shape[2]
        out = np.zeros((self.h_out, self.w_out) if ch == 1 else (self.h_out, self.w_out, ch), np.uint8)
        self.ctx.check(hip.lib.gh_undistort_host(self.h, img.ctypes.data_as(C.c_void_p), ch,
                                                 out.ctypes.data_as(C.c_void_p), 1 if fast else 0))
        return out
