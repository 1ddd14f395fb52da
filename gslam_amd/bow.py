"""Host-side mirror of the Vocabulary plugin (gslam_amd/plugin/vocabulary_plugin.cpp): GSLAM::Vocabulary's BoW
transform on the GPU.  Mirrors Vocabulary::transform(features, bow, fv, levelsup) (GSLAM/core/Vocabulary.h:1558-1621).
torch tensors are device buffers only."""
import ctypes as C

import numpy as np
import torch

from . import hip


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class Vocabulary:
    def __init__(self, ctx: hip.Context, voc: dict):
        """voc: dict(k, L, weighting, scoring, nodes (structured childNum/weight), desc (nnodes x W u8, W a multiple of 8:
        32-byte ORB / BRIEF words, 64-byte long binary descriptors ...))."""
        self.ctx = ctx
        self.k, self.L = int(voc["k"]), int(voc["L"])
        nodes = np.ascontiguousarray(voc["nodes"])
        self.is_float = voc["desc"].dtype == np.float32  # float (L2) vocabulary: desc nnodes x dims float32, dims % 8 == 0
        desc = np.ascontiguousarray(voc["desc"], dtype=np.float32 if self.is_float else np.uint8)
        self.desc_bytes = int(desc.shape[1]) * (4 if self.is_float else 1)
        h = C.c_void_p()
        if self.is_float:
            ctx.check(hip.lib.gh_bow_vocab_create_f32(ctx.h, self.k, self.L, int(voc["weighting"]), int(voc["scoring"]), len(nodes),
                                                      nodes.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p),
                                                      int(desc.shape[1]), C.byref(h)))
        else:
            ctx.check(hip.lib.gh_bow_vocab_create_bytes(ctx.h, self.k, self.L, int(voc["weighting"]), int(voc["scoring"]),
                                                        len(nodes), nodes.ctypes.data_as(C.c_void_p),
                                                        desc.ctypes.data_as(C.c_void_p), self.desc_bytes, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            hip.lib.gh_bow_vocab_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def alloc(self, n_images, cap, device="cuda"):
        z = lambda dt: torch.empty((n_images, cap), dtype=dt, device=device)
        return (z(torch.int32), z(torch.float32), z(torch.int32), z(torch.int32), z(torch.float32),
                torch.empty(n_images, dtype=torch.int32, device=device))

    def transform(self, desc: torch.Tensor, counts=None, levelsup=2, out=None):
        """desc: B x cap x desc_bytes u8 (cuda) -> (word, weight, node, bow_word, bow_val, bow_n) device tensors."""
        B, cap = desc.shape[0], desc.shape[1]
        out = out or self.alloc(B, cap, desc.device)
        self.ctx.check(hip.lib.gh_bow_transform_dev(self.h, _p(desc), _p(counts), cap, B, int(levelsup), *[_p(t) for t in out]))
        return out

    def transform_host(self, desc: np.ndarray, levelsup=2):
        if self.is_float:
            desc = np.ascontiguousarray(desc, dtype=np.float32).reshape(-1, self.desc_bytes // 4)
        else:
            desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, self.desc_bytes)
        n = desc.shape[0]
        m = max(n, 1)
        word, node, bw = np.zeros(m, np.uint32), np.zeros(m, np.uint32), np.zeros(m, np.uint32)
        weight, bv = np.zeros(m, np.float32), np.zeros(m, np.float32)
        nb = C.c_int32()
        pv = lambda a: a.ctypes.data_as(C.c_void_p)
        self.ctx.check(hip.lib.gh_bow_transform_host(self.h, pv(desc), n, int(levelsup), pv(word), pv(weight), pv(node),
                                                     pv(bw), pv(bv), C.byref(nb)))
        return word[:n], weight[:n], node[:n], bw[:nb.value].copy(), bv[:nb.value].copy()


def score(ctx: hip.Context, scoring: int, q, db):
    """All-pairs GSLAM::Vocabulary::score on the GPU.  q / db: (bow_word, bow_val, bow_n) device tensors in the padded
    layout Vocabulary.transform returns (n_q x cap_q, n_db x cap_db).  Returns an n_q x n_db float64 device tensor."""
    qw, qv, qn = q
    dw, dv, dn = db
    out = torch.empty((qw.shape[0], dw.shape[0]), dtype=torch.float64, device=qw.device)
    ctx.check(hip.lib.gh_bow_score_dev(ctx.h, int(scoring), _p(qw), _p(qv), _p(qn), qw.shape[0], qw.shape[1], _p(dw), _p(dv),
                                       _p(dn), dw.shape[0], dw.shape[1], _p(out)))
    return out


def score_host(ctx: hip.Context, scoring: int, q, db_list):
    """One query (ids, vals) against a list of host BowVectors [(ids, vals), ...] -> numpy float64 scores."""
    qi, qv = np.ascontiguousarray(q[0], np.uint32), np.ascontiguousarray(q[1], np.float32)
    off = np.zeros(len(db_list) + 1, np.int64)
    off[1:] = np.cumsum([len(d[0]) for d in db_list])
    di = np.ascontiguousarray(np.concatenate([np.asarray(d[0], np.uint32) for d in db_list]) if db_list else np.zeros(0, np.uint32))
    dv = np.ascontiguousarray(np.concatenate([np.asarray(d[1], np.float32) for d in db_list]) if db_list else np.zeros(0, np.float32))
    out = np.zeros(len(db_list), np.float64)
    pv = lambda a: a.ctypes.data_as(C.c_void_p)
    ctx.check(hip.lib.gh_bow_score_host(ctx.h, int(scoring), pv(qi), pv(qv), len(qi), pv(di), pv(dv), pv(off), len(db_list),
                                        pv(out)))
    return out
