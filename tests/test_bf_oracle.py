"""CPU-side checks of the BF oracle: pinned against the reference's own hamming32 (oracle/_ref,
built from /root/reference/GSLAM/core/Vocabulary.h:485-491) and against committed golden vectors."""
import os

import numpy as np
import pytest

import oracle_lib

GOLD = os.path.join(os.path.dirname(__file__), "golden", "bf_reference.npz")


def test_golden_vectors_from_reference(oracle):
    g = np.load(GOLD)
    idx1, d1, d2 = oracle.bf_match(g["q"], g["t"])
    assert np.array_equal(idx1, g["idx1"])
    assert np.array_equal(d1.astype(np.float32), g["d1"])  # reference distance is float-typed
    # pairwise distance matrix of the reference's hamming32 on a small block
    qs, ts = g["q"][:16], g["t"][:16]
    for i in range(16):
        for j in range(16):
            assert oracle.hamming32(qs[i], ts[j]) == int(g["dmat"][i, j])


@pytest.mark.skipif(not oracle_lib.have_reference(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_equals_reference_live(oracle):
    ref = oracle_lib.load_reference()
    q = oracle_lib.random_descriptors(257, 11)
    t, _ = oracle_lib.correlated_descriptors(oracle_lib.random_descriptors(301, 11)[:301], 5)
    t[7] = t[3]  # exact duplicate rows -> tie must resolve to the lower index
    idx_r, d_r = ref.bf_match(q, t)
    idx_o, d1, d2 = oracle.bf_match(q, t)
    assert np.array_equal(idx_r, idx_o)
    assert np.array_equal(d_r, d1.astype(np.float32))
    assert (d2 >= d1).all()


def test_tie_break_lowest_index(oracle):
    q = oracle_lib.random_descriptors(4, 1)
    t = np.repeat(q[:1], 5, axis=0)  # five identical train rows
    idx1, d1, d2 = oracle.bf_match(q[:1], t)
    assert idx1[0] == 0 and d1[0] == 0 and d2[0] == 0


def test_empty_and_single_train(oracle):
    q = oracle_lib.random_descriptors(3, 2)
    idx1, d1, d2 = oracle.bf_match(q, np.zeros((0, 32), np.uint8))
    assert (idx1 == -1).all() and (d1 == 65535).all() and (d2 == 65535).all()
    idx1, d1, d2 = oracle.bf_match(q, q[:1])
    assert (idx1 == 0).all() and d1[0] == 0 and (d2 == 65535).all()


def test_omp_equals_serial(oracle):
    q = oracle_lib.random_descriptors(100, 3)
    t = oracle_lib.random_descriptors(77, 4)
    a = oracle.bf_match(q, t)
    b = oracle.bf_match(q, t, threads=4)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_mask_semantics(oracle):
    idx1 = np.array([0, 1, 2, -1], np.int32)
    d1 = np.array([10, 60, 30, 65535], np.uint16)
    d2 = np.array([40, 61, 31, 65535], np.uint16)
    back = np.array([0, 1, 0], np.int32)
    keep = oracle.match_mask(idx1, d1, d2, back, 3, 50, 7, 10, 1)
    # row0: 10<=50, 10*10<7*40, back[0]==0 -> keep; row1: d1>50; row2: ratio 300<217 false; row3: no match
    assert keep.tolist() == [1, 0, 0, 0]


# ------------------------------------------------------------------ wider descriptors (hamming64 / hamming8x)
GOLD_BYTES = os.path.join(os.path.dirname(__file__), "golden", "bf_bytes_reference.npz")


def test_wide_descriptor_golden_vectors_from_reference(oracle):
    """oracle_bf_match_bytes against vectors generated from the reference's own hamming64 / hamming8x (tools/gen_golden.py
    bf_bytes; GSLAM/core/Vocabulary.h:493-513,565-567)."""
    g = np.load(GOLD_BYTES)
    for nb in (64, 40, 16, 128):
        idx1, d1, d2 = oracle.bf_match_bytes(g["q%d" % nb], g["t%d" % nb], nb)
        assert np.array_equal(idx1, g["idx%d" % nb]) and np.array_equal(d1.astype(np.float32), g["d%d" % nb])
        assert (d2 >= d1).all()
    # 32 bytes through the generic entry is hamming32
    q, t = oracle_lib.random_descriptors(40, 3), oracle_lib.random_descriptors(50, 4)
    for a, b in zip(oracle.bf_match_bytes(q, t, 32), oracle.bf_match(q, t)):
        assert np.array_equal(a, b)


@pytest.mark.skipif(not oracle_lib.have_reference(), reason="oracle/_ref not built (needs /root/reference)")
def test_wide_descriptor_oracle_equals_reference_live(oracle):
    ref = oracle_lib.load_reference()
    rng = np.random.default_rng(5)
    for nb in (64, 8, 24, 72, 256):
        q = rng.integers(0, 256, size=(70, nb), dtype=np.uint8)
        t = rng.integers(0, 256, size=(90, nb), dtype=np.uint8)
        t[33] = t[2]
        t[:20] = q[:20]
        idx_r, d_r = ref.bf_match_bytes(q, t, nb)
        idx_o, d1, d2 = oracle.bf_match_bytes(q, t, nb)
        assert np.array_equal(idx_r, idx_o) and np.array_equal(d_r, d1.astype(np.float32))
