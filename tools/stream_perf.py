"""Host-fed extraction through gh_orb_stream_* (no torch): throughput by chunk size / depth, and single-frame latency.
   python tools/stream_perf.py [frames]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gslam_amd import hip  # noqa: E402
from gslam_amd.orb import OrbStream  # noqa: E402


def pinned(ctx, nbytes):
    p = C.c_void_p()
    ctx.check(hip.lib.gh_host_alloc_pinned(ctx.h, C.c_size_t(nbytes), C.byref(p)))
    return np.frombuffer((C.c_uint8 * nbytes).from_address(p.value), np.uint8), p


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    W, H, K = 1920, 1080, 2000
    ctx = hip.Context(0)
    host, hp = pinned(ctx, F * W * H)
    # frames: the synthetic generator on the device, downloaded once
    d = C.c_void_p()
    ctx.check(hip.lib.gh_dev_alloc(ctx.h, C.c_size_t(F * W * H), C.byref(d)))
    ctx.check(hip.lib.gh_synth_frames_dev(ctx.h, d, W, H, W, C.c_size_t(W * H), 0, F, C.c_uint32(0x5EED0000)))
    ctx.check(hip.lib.gh_dev_download(ctx.h, hp, d, C.c_size_t(F * W * H)))
    ctx.check(hip.lib.gh_dev_free(ctx.h, d))
    frames = host.reshape(F, W * H)
    for chunk, depth in ((50, 3), (25, 3), (10, 4), (100, 2), (50, 2), (4, 4), (1, 4)):
        st = OrbStream(ctx, W, H, chunk, depth, n_features=K)
        n_chunks = F // chunk

        def run():
            tickets, total = [], 0
            for c in range(n_chunks):
                tickets.append(st.submit(frames[c * chunk:(c + 1) * chunk]))
                if len(tickets) >= depth:
                    off, _, _, _ = st.collect(tickets.pop(0), copy=False)
                    total += int(off[-1])
            for t in tickets:
                off, _, _, _ = st.collect(t, copy=False)
                total += int(off[-1])
            return total
        run()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            kp = run()
        dt = (time.perf_counter() - t0) / reps
        nf = n_chunks * chunk
        print(f"chunk {chunk:4d} depth {depth}: {dt * 1e3:8.2f} ms for {nf} frames -> {kp / dt / 1e6:6.1f} Mkeypoints/s, "
              f"{nf * W * H / dt / 1e9:5.1f} GB/s of frames over PCIe, {dt / nf * 1e6:6.1f} us/frame", flush=True)
        st.close()
    # single-frame latency: submit + collect, nothing else in flight
    for (w, h, k) in ((640, 480, 1000), (1920, 1080, 2000)):
        st = OrbStream(ctx, w, h, 1, 1, n_features=k)
        fr = np.ascontiguousarray(frames[0].reshape(H, W)[:h, :w]).reshape(1, -1)
        buf = st.staging()
        lat, gpu = [], []
        for i in range(300):
            t0 = time.perf_counter()
            buf[0, : w * h] = fr[0]
            t = st.submit(None, 1)
            _, _, _, g = st.collect(t, copy=False)
            lat.append((time.perf_counter() - t0) * 1e3)
            gpu.append(g)
        lat, gpu = np.sort(lat[50:]), np.sort(gpu[50:])
        print(f"latency {w}x{h} K={k}: host p50 {lat[len(lat) // 2]:.3f} ms p99 {lat[int(len(lat) * 0.99)]:.3f} ms; "
              f"gpu (link in -> results out) p50 {gpu[len(gpu) // 2]:.3f} ms", flush=True)
        st.close()


if __name__ == "__main__":
    main()
