"""Host-side mirror of the extractor half of the FeatureDetector plugin interface
(gslam_amd/plugin/FeatureDetector.h) for tests and bench.py: batched ORB extraction on frames resident
in HBM.  torch tensors are device buffers only; all compute is libgslam_hip.so.

Output layout follows GSLAM/core/Map.h:122-195 (KeyPoint, 28 B) and Map.h:309-321 (descriptors as an
N x 32 8UC1 matrix).
"""
import ctypes as C

import numpy as np
import torch

from . import hip

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def synth_frames(ctx, n_frames, width, height, base_seed=0x5EED0000, first_frame=0, row_stride=None,
                 device="cuda"):
    """n_frames deterministic gray frames generated in HBM (bit-identical to oracle/synth.c)."""
    row_stride = row_stride or width
    out = torch.empty((n_frames, height, row_stride), dtype=torch.uint8, device=device)
    ctx.check(hip.lib.gh_synth_frames_dev(ctx.h, _p(out), width, height, row_stride,
                                          C.c_size_t(height * row_stride), int(first_frame), int(n_frames),
                                          C.c_uint32(base_seed & 0xFFFFFFFF)))
    return out


class OrbExtractor:
    def __init__(self, ctx: hip.Context, width, height, max_batch=1, n_features=1000, n_levels=8, ini_th=20,
                 min_th=7):
        self.ctx = ctx
        self.w, self.h, self.max_batch, self.K, self.L = width, height, max_batch, n_features, n_levels
        prm = hip.OrbParams(n_features, n_levels, ini_th, min_th)
        h = C.c_void_p()
        ctx.check(hip.lib.gh_orb_plan_create(ctx.h, width, height, max_batch, C.byref(prm), C.byref(h)))
        self.plan = h

    def close(self):
        if self.plan:
            hip.lib.gh_orb_plan_destroy(self.plan)
            self.plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_pattern(self, pattern):
        """pattern: 256 x 4 int8 (ax, ay, bx, by), unrotated.  See gh_orb_plan_set_pattern."""
        pat = np.ascontiguousarray(pattern, dtype=np.int8).reshape(256, 4)
        self.ctx.check(hip.lib.gh_orb_plan_set_pattern(self.plan, pat.ctypes.data_as(C.c_void_p)))

    def set_steering(self, mode):
        """0: 30 orientation bins (default); 1: continuous (fastAtan2 angle, per-keypoint pattern rotation).  See
        gh_orb_plan_set_steering."""
        self.ctx.check(hip.lib.gh_orb_plan_set_steering(self.plan, int(mode)))

    def set_distribution(self, mode):
        """0: 32 x 32 cells + rank order (default); 1: ORB-SLAM's cells + quadtree.  See gh_orb_plan_set_distribution."""
        self.ctx.check(hip.lib.gh_orb_plan_set_distribution(self.plan, int(mode)))

    def level(self, l):
        w, h, q = C.c_int(), C.c_int(), C.c_int()
        self.ctx.check(hip.lib.gh_orb_plan_level(self.plan, l, C.byref(w), C.byref(h), C.byref(q)))
        return w.value, h.value, q.value

    def device_bytes(self):
        return int(hip.lib.gh_orb_plan_device_bytes(self.plan))

    def alloc_outputs(self, batch, device="cuda"):
        kps = torch.empty((batch, self.K, 7), dtype=torch.float32, device=device)  # 28-byte records
        desc = torch.empty((batch, self.K, 32), dtype=torch.uint8, device=device)
        counts = torch.empty(batch, dtype=torch.int32, device=device)
        return kps, desc, counts

    def extract(self, frames: torch.Tensor, out=None):
        """frames: B x H x stride u8 (cuda).  Returns (kps B x K x 7 f32-view of KeyPoint, desc B x K x 32, counts).
        Small calls (up to two 1080p frames) replay a captured launch graph when the SAME frames / out buffers come back
        (pass out= and reuse it); with fresh buffers every call the plan stops capturing after a few misses."""
        assert frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 3
        B, H, stride = frames.shape
        assert H == self.h and stride >= self.w and frames.stride(2) == 1 and frames.stride(1) == stride
        if out is None:
            out = self.alloc_outputs(B, frames.device)
        kps, desc, counts = out
        self.ctx.check(hip.lib.gh_orb_extract_dev(self.plan, _p(frames), B, C.c_size_t(frames.stride(0)), stride,
                                                  _p(kps), _p(desc), _p(counts)))
        return kps, desc, counts

    def extract_host(self, gray: np.ndarray):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        assert gray.shape == (self.h, self.w)
        kps = np.zeros(self.K, KP_DTYPE)
        desc = np.zeros((self.K, 32), np.uint8)
        n = C.c_int32()
        self.ctx.check(hip.lib.gh_orb_extract_host(self.plan, gray.ctypes.data_as(C.c_void_p), self.w,
                                                   kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p),
                                                   C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    DBG_NAMES = ("cells", "dense_cells", "overflow_cells", "cap_cells", "rank_dropped", "strong_silenced",
                 "max_queue", "max_nz", "sel_cut", "sel_tie_split", "sel_overflow_cells", "sel_streamed",
                 "unused_slots", "starved_levels", "weak_cells", "resize_passes")

    def debug_counters(self, enable=True, read=True):
        """Branch census of the extractions since the last read (gh_orb_plan_debug_counters)."""
        out = np.zeros(16, np.uint32)
        self.ctx.check(hip.lib.gh_orb_plan_debug_counters(self.plan, 1 if enable else 0,
                                                          out.ctypes.data_as(C.c_void_p) if read else None))
        return dict(zip(self.DBG_NAMES, out.tolist()))

    def debug_level(self, slot, level):
        w, h, _ = self.level(level)
        out = np.zeros((h, w), np.uint8)
        self.ctx.check(hip.lib.gh_orb_debug_level(self.plan, slot, level, out.ctypes.data_as(C.c_void_p)))
        return out


def kps_to_numpy(kps_tensor, counts=None):
    """B x K x 7 float32 tensor -> structured KeyPoint array (bit reinterpretation, no conversion)."""
    a = kps_tensor.cpu().numpy()
    return a.view(KP_DTYPE).reshape(a.shape[0], a.shape[1])


class OrbStream:
    """gh_orb_stream_*: host-fed extraction (frames in host memory -> packed records in host memory) on three HIP streams.
    No torch involved.  Mirrors the C ABI one to one; numpy views of the pinned blocks are handed out."""

    def __init__(self, ctx: hip.Context, width, height, chunk_frames, depth=3, channels=1, row_stride=None,
                 frame_stride=None, n_features=1000, n_levels=8, ini_th=20, min_th=7):
        self.ctx = ctx
        self.w, self.h, self.ch = width, height, channels
        self.row_stride = row_stride or width * channels
        self.frame_stride = frame_stride or self.row_stride * height
        self.chunk, self.depth, self.K = chunk_frames, depth, n_features
        prm = hip.OrbParams(n_features, n_levels, ini_th, min_th)
        h = C.c_void_p()
        ctx.check(hip.lib.gh_orb_stream_create(ctx.h, width, height, channels, self.row_stride,
                                               C.c_size_t(self.frame_stride), chunk_frames, depth, C.byref(prm),
                                               C.byref(h)))
        self.s = h

    def close(self):
        if self.s:
            hip.lib.gh_orb_stream_destroy(self.s)
            self.s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def staging(self):
        """numpy view (chunk x frame_stride bytes) of the pinned staging block the next submit will use."""
        p = C.c_void_p()
        self.ctx.check(hip.lib.gh_orb_stream_staging(self.s, C.byref(p)))
        buf = (C.c_uint8 * (self.chunk * self.frame_stride)).from_address(p.value)
        return np.frombuffer(buf, np.uint8).reshape(self.chunk, self.frame_stride)

    def submit(self, frames=None, n_frames=None):
        """frames: None (the staging block was filled) or a C-contiguous uint8 array laid out as the stream expects."""
        t = C.c_int64()
        if frames is None:
            self.ctx.check(hip.lib.gh_orb_stream_submit(self.s, None, int(n_frames), C.byref(t)))
        else:
            assert frames.dtype == np.uint8 and frames.flags.c_contiguous
            n = int(n_frames if n_frames is not None else frames.shape[0])
            self.ctx.check(hip.lib.gh_orb_stream_submit(self.s, frames.ctypes.data_as(C.c_void_p), n, C.byref(t)))
        return t.value

    def poll(self, ticket):
        r = C.c_int()
        self.ctx.check(hip.lib.gh_orb_stream_poll(self.s, C.c_int64(ticket), C.byref(r)))
        return bool(r.value)

    def collect(self, ticket, copy=True):
        """-> (offsets int32[n + 1], kps KP_DTYPE[total], desc uint8[total, 32], gpu_ms)."""
        r = hip.OrbStreamResult()
        self.ctx.check(hip.lib.gh_orb_stream_collect(self.s, C.c_int64(ticket), C.byref(r)))
        n = r.n_frames
        off = np.ctypeslib.as_array(r.offsets, shape=(n + 1,))
        total = int(off[n])
        kps = np.frombuffer((C.c_uint8 * (total * 28)).from_address(r.kps), KP_DTYPE) if total else np.zeros(0, KP_DTYPE)
        desc = (np.frombuffer((C.c_uint8 * (total * 32)).from_address(r.desc), np.uint8).reshape(total, 32) if total
                else np.zeros((0, 32), np.uint8))
        if copy:
            off, kps, desc = off.copy(), kps.copy(), desc.copy()
        return off, kps, desc, float(r.gpu_ms)
