"""ctypes loader for oracle/liboracle.so (and oracle/_ref when present).  Only tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() may import this."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libgslam_ref.so")
REF_POPCNT_SO = os.path.join(ROOT, "oracle", "_ref", "libgslam_ref_popcnt.so")

_vp = C.c_void_p


def _ptr(a):
    return a.ctypes.data_as(_vp) if a is not None else None


class Oracle:
    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make oracle`")
        self.lib = C.CDLL(path)

    # ---------------------------------------------------------------- BF matcher
    def hamming32(self, a, b):
        return int(self.lib.oracle_hamming32(_ptr(np.ascontiguousarray(a)), _ptr(np.ascontiguousarray(b))))

    def bf_match(self, q, t, threads=1):
        q = np.ascontiguousarray(q, dtype=np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, dtype=np.uint8).reshape(-1, 32)
        nq, nt = q.shape[0], t.shape[0]
        idx1 = np.empty(nq, np.int32)
        d1 = np.empty(nq, np.uint16)
        d2 = np.empty(nq, np.uint16)
        if threads == 1:
            self.lib.oracle_bf_match(_ptr(q), nq, _ptr(t), nt, _ptr(idx1), _ptr(d1), _ptr(d2))
        else:
            self.lib.oracle_bf_match_omp(_ptr(q), nq, _ptr(t), nt, _ptr(idx1), _ptr(d1), _ptr(d2), int(threads))
        return idx1, d1, d2

    def match_mask(self, idx1, d1, d2, back, nt, max_dist, ratio_num, ratio_den, cross_check):
        nq = idx1.shape[0]
        keep = np.empty(nq, np.uint8)
        self.lib.oracle_match_mask(_ptr(idx1), _ptr(d1), _ptr(d2), nq, _ptr(back), int(nt), int(max_dist),
                                   int(ratio_num), int(ratio_den), int(cross_check), _ptr(keep))
        return keep


class Reference:
    """The reference's own code compiled from /root/reference (oracle/_ref)."""

    def __init__(self, path=REF_SO):
        self.lib = C.CDLL(path)
        self.lib.ref_hamming32.restype = C.c_float

    def hamming32(self, a, b):
        return float(self.lib.ref_hamming32(_ptr(a), _ptr(b)))

    def bf_match(self, q, t, threads=1):
        q = np.ascontiguousarray(q, dtype=np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, dtype=np.uint8).reshape(-1, 32)
        idx1 = np.empty(q.shape[0], np.int32)
        d1 = np.empty(q.shape[0], np.float32)
        self.lib.ref_bf_match(_ptr(q), q.shape[0], _ptr(t), t.shape[0], _ptr(idx1), _ptr(d1), int(threads))
        return idx1, d1

    def _se3(self, name, *ins, out_n):
        out = np.zeros(out_n, np.float64)
        args = [_ptr(np.ascontiguousarray(a, dtype=np.float64)) for a in ins]
        getattr(self.lib, name)(*args, _ptr(out))
        return out

    def se3_exp(self, xi):
        return self._se3("ref_se3_exp", xi, out_n=7)

    def se3_log(self, pose):
        return self._se3("ref_se3_log", pose, out_n=6)

    def se3_mul(self, a, b):
        return self._se3("ref_se3_mul", a, b, out_n=7)

    def se3_inverse(self, a):
        return self._se3("ref_se3_inverse", a, out_n=7)

    def se3_apply(self, a, p):
        return self._se3("ref_se3_apply", a, p, out_n=3)


def load():
    return Oracle()


def have_reference():
    return os.path.exists(REF_SO)


def load_reference(popcnt=False):
    return Reference(REF_POPCNT_SO if popcnt else REF_SO)


# ---------------------------------------------------------------- deterministic inputs
def splitmix64(seed, n):
    """n uint64 values of the splitmix64 stream started at `seed` (vectorised)."""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def random_descriptors(n, seed):
    return splitmix64(seed, n * 4).view(np.uint8).reshape(n, 32).copy()


def correlated_descriptors(base, seed, flip_frac=0.10, replace_frac=0.20):
    """frame t+1 = frame t with ~10% of bits flipped and ~20% of rows replaced."""
    n = base.shape[0]
    r = splitmix64(seed, n * 32 * 8 + n + n * 4)
    bits = (r[: n * 256] % np.uint64(1000)) < np.uint64(int(flip_frac * 1000))
    flip = np.packbits(bits.reshape(n, 256), axis=1, bitorder="little")
    out = base ^ flip
    rows = (r[n * 256: n * 256 + n] % np.uint64(1000)) < np.uint64(int(replace_frac * 1000))
    fresh = r[n * 256 + n:].view(np.uint8).reshape(n, 32)
    out[rows] = fresh[rows]
    # shuffle rows deterministically so indices are not the identity
    perm = np.argsort(splitmix64(seed ^ 0xABCDEF, n), kind="stable")
    return out[perm].copy(), perm
