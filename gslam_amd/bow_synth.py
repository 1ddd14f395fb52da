"""Synthetic DBoW-style vocabularies in GSLAM's binary .gbow layout (GSLAM/core/Vocabulary.h:1843-1932):
u64 magic 88877711233, u8 compressed(0), u32 nnodes, i32 k, L, scoring, weighting, cols(32), rows(1), type(0),
Node{u32 childNum; f32 weight}[nnodes], nnodes x 32 descriptor bytes.  Children of node p live at p*k+1 .. p*k+childNum
(Vocabulary.h:1716).  No vocabulary file exists offline and the reference's trainer is unseeded/buggy,
so tests and bench build one here: child descriptors are their parent's with a few bits flipped (so the greedy Hamming
descent is meaningful), leaf weights are IDF-like positive floats with a few zeros ("stopped" words) and a few internal
nodes carry fewer than k children.  numpy only."""
import struct

import numpy as np

MAGIC = 88877711233
TF_IDF, TF, IDF, BINARY = 0, 1, 2, 3
L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT = 0, 1, 2, 3, 4, 5


def make_vocabulary(k=10, L=4, seed=1, weighting=TF_IDF, scoring=L1_NORM, ragged=True, stop_frac=0.02, desc_bytes=32):
    rng = np.random.default_rng(seed)
    nnodes = (k ** (L + 1) - 1) // (k - 1)
    desc = np.zeros((nnodes, desc_bytes), np.uint8)  # any multiple of 8 bytes (Vocabulary.h:560-568)
    child_num = np.zeros(nnodes, np.uint32)
    weight = np.zeros(nnodes, np.float32)
    desc[0] = rng.integers(0, 256, desc_bytes, dtype=np.uint8)
    level_start = 0
    for lvl in range(L):
        n_lvl = k ** lvl
        parents = np.arange(level_start, level_start + n_lvl)
        child_num[parents] = k
        if ragged and lvl >= 1:
            short = parents[rng.random(n_lvl) < 0.05]
            child_num[short] = rng.integers(1, k, len(short))
        # children = parent with ~ (48 >> lvl) + 8 random bits flipped
        nflip = max(8, 48 >> lvl)
        kids = parents[:, None] * k + 1 + np.arange(k)[None, :]
        base = np.repeat(desc[parents], k, axis=0)
        flips = np.zeros((n_lvl * k, 8 * desc_bytes), np.uint8)
        cols = rng.integers(0, 8 * desc_bytes, (n_lvl * k, nflip))
        np.put_along_axis(flips, cols, 1, axis=1)
        desc[kids.reshape(-1)] = base ^ np.packbits(flips, axis=1, bitorder="little")
        level_start += n_lvl
    leaves = child_num == 0
    # nodes below a short parent's used children are unreachable; leaves at depth L and "short" subtrees' ends
    weight[leaves] = rng.uniform(0.5, 9.0, int(leaves.sum())).astype(np.float32)
    weight[leaves & (rng.random(nnodes) < stop_frac)] = 0.0
    nodes = np.zeros(nnodes, dtype=[("childNum", "<u4"), ("weight", "<f4")])
    nodes["childNum"] = child_num
    nodes["weight"] = weight
    return {"k": k, "L": L, "weighting": weighting, "scoring": scoring, "nodes": nodes, "desc": desc}


def to_gbow_bytes(v):
    hdr = struct.pack("<QBI", MAGIC, 0, len(v["nodes"]))
    is_float = v["desc"].dtype == np.float32  # GElementType_32F = 5, GElementType_8U = 0 (GSLAM/core/GImage.h:60-69)
    hdr += struct.pack("<7i", v["k"], v["L"], v["scoring"], v["weighting"], v["desc"].shape[1], 1, 5 if is_float else 0)
    return hdr + v["nodes"].tobytes() + v["desc"].tobytes()


def make_float_vocabulary(k=8, L=3, dims=64, seed=1, weighting=TF_IDF, scoring=L1_NORM, stop_frac=0.02):
    """A float (L2) vocabulary in the style of SIFT / SURF trees: every node's descriptor is its parent's plus noise that
    shrinks with the level.  desc: nnodes x dims float32 (dims a multiple of 8)."""
    rng = np.random.default_rng(seed)
    nnodes = (k ** (L + 1) - 1) // (k - 1)
    desc = np.zeros((nnodes, dims), np.float32)
    child_num = np.zeros(nnodes, np.uint32)
    weight = np.zeros(nnodes, np.float32)
    desc[0] = rng.normal(size=dims).astype(np.float32)
    level_start = 0
    for lvl in range(L):
        n_lvl = k ** lvl
        parents = np.arange(level_start, level_start + n_lvl)
        child_num[parents] = k
        kids = parents[:, None] * k + 1 + np.arange(k)[None, :]
        base = np.repeat(desc[parents], k, axis=0)
        desc[kids.reshape(-1)] = base + (rng.normal(size=base.shape) * (1.0 / (1 << lvl))).astype(np.float32)
        level_start += n_lvl
    leaves = child_num == 0
    weight[leaves] = rng.uniform(0.5, 9.0, int(leaves.sum())).astype(np.float32)
    weight[leaves & (rng.random(nnodes) < stop_frac)] = 0.0
    nodes = np.zeros(nnodes, dtype=[("childNum", "<u4"), ("weight", "<f4")])
    nodes["childNum"] = child_num
    nodes["weight"] = weight
    return {"k": k, "L": L, "weighting": weighting, "scoring": scoring, "nodes": nodes, "desc": desc}


def float_features_near_words(v, n, seed=2, sigma=0.05):
    rng = np.random.default_rng(seed)
    ids = rng.integers(len(v["nodes"]) // 2, len(v["nodes"]), n)
    return (v["desc"][ids] + rng.normal(size=(n, v["desc"].shape[1])) * sigma).astype(np.float32)


def features_near_words(v, n, seed=2, flip_bits=10):
    """n descriptors = random leaf-ish node descriptors with a few bits flipped."""
    rng = np.random.default_rng(seed)
    ids = rng.integers(len(v["nodes"]) // 2, len(v["nodes"]), n)
    d = v["desc"][ids].copy()
    flips = np.zeros((n, 8 * d.shape[1]), np.uint8)
    cols = rng.integers(0, 8 * d.shape[1], (n, flip_bits))
    np.put_along_axis(flips, cols, 1, axis=1)
    return d ^ np.packbits(flips, axis=1, bitorder="little")
