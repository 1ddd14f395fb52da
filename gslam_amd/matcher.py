"""Host-side mirror of the matcher half of the FeatureDetector plugin interface
(gslam_amd/plugin/FeatureDetector.h), for tests and bench.py.  torch tensors are only the device
buffers; all compute is libgslam_hip.so.

Reference anchors: GSLAM/core/Vocabulary.h:485-491 (distance), :1712-1725 (first-minimum rule),
GSLAM/core/Map.h:252-258 (match list = vector<pair<int,int>>).
"""
import ctypes as C

import torch

from . import hip


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class BFMatcher:
    def __init__(self, ctx: hip.Context):
        self.ctx = ctx

    def match(self, q: torch.Tensor, t: torch.Tensor):
        """q: nq x 32 u8 (cuda), t: nt x 32 u8 (cuda) -> idx1 int32, d1 u16, d2 u16 (as int16 views)."""
        assert q.is_cuda and q.dtype == torch.uint8 and q.is_contiguous()
        assert t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()
        nq, nt = q.shape[0], t.shape[0]
        idx1 = torch.empty(nq, dtype=torch.int32, device=q.device)
        d1 = torch.empty(nq, dtype=torch.int16, device=q.device)
        d2 = torch.empty(nq, dtype=torch.int16, device=q.device)
        self.ctx.check(hip.lib.gh_bf_match_dev(self.ctx.h, _p(q), nq, _p(t), nt, _p(idx1), _p(d1), _p(d2)))
        return idx1, d1, d2

    def match_bytes(self, q: torch.Tensor, t: torch.Tensor):
        """Descriptors of any width that is a multiple of 8 bytes (q: nq x B u8, t: nt x B u8): gh_bf_match_bytes_dev --
        hamming64 / hamming8x of the reference (GSLAM/core/Vocabulary.h:493-513), same outputs as match()."""
        assert q.is_cuda and t.is_cuda and q.dtype == t.dtype == torch.uint8 and q.is_contiguous() and t.is_contiguous()
        nq, nt, nb = q.shape[0], t.shape[0], q.shape[1]
        assert t.shape[1] == nb or nt == 0
        idx1 = torch.empty(nq, dtype=torch.int32, device=q.device)
        d1 = torch.empty(nq, dtype=torch.int16, device=q.device)
        d2 = torch.empty(nq, dtype=torch.int16, device=q.device)
        self.ctx.check(hip.lib.gh_bf_match_bytes_dev(self.ctx.h, _p(q), nq, _p(t), nt, nb, _p(idx1), _p(d1), _p(d2)))
        return idx1, d1, d2

    def match_pairs_bytes(self, desc: torch.Tensor, counts: torch.Tensor, pair_q: torch.Tensor, pair_t: torch.Tensor):
        """desc: F x cap x B u8 (B a multiple of 8): gh_bf_match_pairs_bytes_dev."""
        cap, nb, P = desc.shape[1], desc.shape[2], pair_q.shape[0]
        idx1 = torch.empty((P, cap), dtype=torch.int32, device=desc.device)
        d1 = torch.empty((P, cap), dtype=torch.int16, device=desc.device)
        d2 = torch.empty((P, cap), dtype=torch.int16, device=desc.device)
        self.ctx.check(hip.lib.gh_bf_match_pairs_bytes_dev(self.ctx.h, _p(desc), _p(counts), cap, nb, _p(pair_q), _p(pair_t), P,
                                                           _p(idx1), _p(d1), _p(d2)))
        return idx1, d1, d2

    def match_pairs(self, desc: torch.Tensor, counts: torch.Tensor, pair_q: torch.Tensor, pair_t: torch.Tensor,
                    out=None, mfma=None):
        """desc: F x cap x 32 u8; counts: F int32; pair_q/pair_t: P int32 -> (P x cap) idx1, d1, d2.
        mfma=None: gh_bf_match_pairs_dev (dispatches on the amount of pair work); True: the exact integer MFMA formulation
        (gh_bf_match_pairs_mfma_dev); False: the popcount kernel (gh_bf_match_pairs_popc_dev).  Same results all three."""
        F, cap = desc.shape[0], desc.shape[1]
        P = pair_q.shape[0]
        if out is None:
            idx1 = torch.empty((P, cap), dtype=torch.int32, device=desc.device)
            d1 = torch.empty((P, cap), dtype=torch.int16, device=desc.device)
            d2 = torch.empty((P, cap), dtype=torch.int16, device=desc.device)
        else:
            idx1, d1, d2 = out
        fn = hip.lib.gh_bf_match_pairs_dev if mfma is None else (
            hip.lib.gh_bf_match_pairs_mfma_dev if mfma else hip.lib.gh_bf_match_pairs_popc_dev)
        self.ctx.check(fn(self.ctx.h, _p(desc), _p(counts), cap, _p(pair_q), _p(pair_t), P, _p(idx1), _p(d1), _p(d2)))
        return idx1, d1, d2

    def match_band_pairs(self, desc, kps, counts, pair_q, pair_t, band_per_size):
        """Stereo variant: kps is the F x cap x 7 float32 view of the KeyPoint records parallel to desc."""
        P, cap = pair_q.shape[0], desc.shape[1]
        idx1 = torch.empty((P, cap), dtype=torch.int32, device=desc.device)
        d1 = torch.empty((P, cap), dtype=torch.int16, device=desc.device)
        d2 = torch.empty((P, cap), dtype=torch.int16, device=desc.device)
        self.ctx.check(hip.lib.gh_bf_match_band_pairs_dev(self.ctx.h, _p(desc), _p(kps), _p(counts), cap, _p(pair_q),
                                                          _p(pair_t), P, C.c_float(band_per_size), _p(idx1), _p(d1),
                                                          _p(d2)))
        return idx1, d1, d2

    def mask(self, idx1, d1, d2, back_idx1=None, nt=0, max_dist=50, ratio_num=0, ratio_den=1, cross_check=False):
        nq = idx1.shape[0]
        keep = torch.empty(nq, dtype=torch.uint8, device=idx1.device)
        self.ctx.check(hip.lib.gh_match_mask_dev(self.ctx.h, _p(idx1), _p(d1), _p(d2), nq, _p(back_idx1), int(nt),
                                                 int(max_dist), int(ratio_num), int(ratio_den),
                                                 1 if cross_check else 0, _p(keep)))
        return keep

    def valu_probe(self):
        r = C.c_double()
        self.ctx.check(hip.lib.gh_bf_valu_probe(self.ctx.h, C.byref(r)))
        return r.value
