"""Adversarial gray images for the ORB parity tests (GPU extractor vs oracle, and oracle vs the independent CPU
restatements).  Every generator is seeded and pure numpy; what each one is meant to reach inside
gslam_amd/csrc/orb.hip is stated next to it (tests/test_orb_adversarial_gpu.py asserts, through
gh_orb_plan_debug_counters, that the branches were really taken).
"""
import numpy as np


def noise(w, h, seed, lo=0, hi=256):
    """Uniform noise: nearly every pixel passes the compass test and scores, every cell holds > 64 scored pixels
    (list branch of fast_cells), > 32 NMS maxima (rank drop, cap 32, overflow slots, select's overflow loads)."""
    return np.random.default_rng(seed).integers(lo, hi, (h, w), dtype=np.uint8)


def binary_noise(w, h, seed, p=0.5):
    """0 / 255 only: saturated differences (score 255 is the largest representable), massive score ties."""
    return np.where(np.random.default_rng(seed).random((h, w)) < p, 255, 0).astype(np.uint8)


def low_contrast_noise(w, h, seed, centre=100, amp=9):
    """Differences around the minimum threshold (7) and far below the initial one (20): the 20 -> 7 fallback decides
    every cell, scores sit in 8 .. 18 so ties are everywhere."""
    return (centre + np.random.default_rng(seed).integers(-amp, amp + 1, (h, w))).astype(np.uint8)


def checkerboard(w, h, period, lo=0, hi=255, phase=(0, 0)):
    """period-px squares of lo / hi.  1 px: no 9-arc exists anywhere (ring alternates); 2 px and up: corners on a
    lattice, all with the SAME score -> quota cuts split a single histogram bin, cell order decides."""
    y, x = np.mgrid[0:h, 0:w]
    return np.where((((x + phase[0]) // period) + ((y + phase[1]) // period)) % 2 == 0, lo, hi).astype(np.uint8)


def step_edges(w, h, seed):
    """0 / 255 axis-parallel and diagonal step edges plus rectangles whose corners sit at, just inside and just outside
    the 19-px border of EVERY pyramid level's valid region (the border tiles' trim masks)."""
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w), np.uint8)
    img[:, w // 2:] = 255
    img[h // 3: h // 3 + 40, :] ^= 255
    yy, xx = np.mgrid[0:h, 0:w]
    img[(xx + yy) % 97 < 11] ^= 255
    for k in range(40):  # blobs hugging the border: corners at distance 17 .. 21 from an image side
        d = 17 + k % 5
        side = k % 4
        s = int(rng.integers(3, 12))
        if side == 0:
            x0, y0 = d, int(rng.integers(0, h - s))
        elif side == 1:
            x0, y0 = w - d - s, int(rng.integers(0, h - s))
        elif side == 2:
            x0, y0 = int(rng.integers(0, w - s)), d
        else:
            x0, y0 = int(rng.integers(0, w - s)), h - d - s
        x0, y0 = max(0, x0), max(0, y0)
        img[y0:y0 + s, x0:x0 + s] = 255 if k % 2 else 0
    return img


def dot_lattice(w, h, pitch, value=255, bg=0, size=1, offset=(19, 19)):
    """Isolated identical dots: each one is a corner with the same score; pitch 2 packs 256 strict maxima into a
    cell is impossible for FAST (the ring of one dot holds its neighbours), pitch >= 8 gives 16 identical-score
    candidates per cell -> ranks 0..15, equal scores everywhere, quota ties between cells."""
    img = np.full((h, w), bg, np.uint8)
    for y in range(offset[1], h - size, pitch):
        for x in range(offset[0], w - size, pitch):
            img[y:y + size, x:x + size] = value
    return img


def few_corners(w, h, n, seed):
    """n isolated bright squares on a flat background: every level is below its quota (starved levels, zero-filled
    output tail)."""
    rng = np.random.default_rng(seed)
    img = np.full((h, w), 60, np.uint8)
    for _ in range(n):
        x = int(rng.integers(min(25, w // 2), max(w - 30, w // 2 + 1)))
        y = int(rng.integers(min(25, h // 2), max(h - 30, h // 2 + 1)))
        img[y:y + 5, x:x + 5] = 200
    return img


def gradient_with_noise_patch(w, h, seed):
    """Smooth ramp (no corners) with one noise patch: a handful of saturated cells, everything else empty."""
    y, x = np.mgrid[0:h, 0:w]
    img = ((x * 255) // max(1, w - 1)).astype(np.uint8)
    pw, ph = min(w, 96), min(h, 80)
    x0, y0 = (w - pw) // 2, (h - ph) // 2
    img[y0:y0 + ph, x0:x0 + pw] = np.random.default_rng(seed).integers(0, 256, (ph, pw), dtype=np.uint8)
    return img


def mixed(w, h, seed):
    """Quadrants of different classes in one frame."""
    img = noise(w, h, seed)
    img[: h // 2, : w // 2] = checkerboard(w // 2, h // 2, 3)
    img[h // 2:, : w // 2] = low_contrast_noise(w // 2, h - h // 2, seed + 1)
    img[: h // 2, w // 2:] = binary_noise(w - w // 2, h // 2, seed + 2, p=0.3)
    return img


CLASSES = {
    "noise": lambda w, h, s: noise(w, h, s),
    "binary_noise": lambda w, h, s: binary_noise(w, h, s),
    "sparse_binary": lambda w, h, s: binary_noise(w, h, s, p=0.03),
    "low_contrast": lambda w, h, s: low_contrast_noise(w, h, s),
    "checker1": lambda w, h, s: checkerboard(w, h, 1),
    "checker2": lambda w, h, s: checkerboard(w, h, 2, phase=(s % 2, (s // 2) % 2)),
    "checker5": lambda w, h, s: checkerboard(w, h, 5, lo=10, hi=245),
    "step_edges": lambda w, h, s: step_edges(w, h, s),
    "dots8": lambda w, h, s: dot_lattice(w, h, 8),
    "dots11": lambda w, h, s: dot_lattice(w, h, 11, value=0, bg=255, size=2, offset=(17, 21)),
    "few_corners": lambda w, h, s: few_corners(w, h, 12, s),
    "ramp_patch": lambda w, h, s: gradient_with_noise_patch(w, h, s),
    "mixed": lambda w, h, s: mixed(w, h, s),
    "white": lambda w, h, s: np.full((h, w), 255, np.uint8),
    "black": lambda w, h, s: np.zeros((h, w), np.uint8),
}
