"""Single-frame latency of the host-buffer entry points (what a per-frame tracking thread pays): gh_orb_extract_host
and gh_bf_match_host, pageable host memory in and out.  Perf probe, not part of the product."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gslam_amd import hip  # noqa: E402
from gslam_amd.orb import OrbExtractor, synth_frames  # noqa: E402


def main():
    ctx = hip.Context(0)
    for (w, h, k) in ((640, 480, 1000), (1241, 376, 2000), (1920, 1080, 2000), (3840, 2160, 2000)):
        ex = OrbExtractor(ctx, w, h, max_batch=1, n_features=k)
        fr = synth_frames(ctx, 2, w, h).cpu().numpy()
        a = ex.extract_host(fr[0])
        b = ex.extract_host(fr[1])
        n = 50
        t = time.perf_counter()
        for i in range(n):
            ex.extract_host(fr[i & 1])
        te = (time.perf_counter() - t) / n
        import ctypes as C
        q, tr = np.ascontiguousarray(a[1]), np.ascontiguousarray(b[1])
        idx = np.zeros(len(q), np.int32)
        d1 = np.zeros(len(q), np.uint16)
        d2 = np.zeros(len(q), np.uint16)

        def match_host():
            ctx.check(hip.lib.gh_bf_match_host(ctx.h, q.ctypes.data_as(C.c_void_p), len(q), tr.ctypes.data_as(C.c_void_p),
                                               len(tr), idx.ctypes.data_as(C.c_void_p), d1.ctypes.data_as(C.c_void_p),
                                               d2.ctypes.data_as(C.c_void_p)))
        match_host()
        t = time.perf_counter()
        for i in range(n):
            match_host()
        tm = (time.perf_counter() - t) / n
        print(f"{w}x{h} K={k}: extract_host {te * 1e6:.0f} us/frame ({len(a[0])} kpts), match_host {tm * 1e6:.0f} us "
              f"({len(a[0])} x {len(b[0])})")
        ex.close()


if __name__ == "__main__":
    main()
