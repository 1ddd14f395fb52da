#!/usr/bin/env python3
"""Generate tests/golden/* from the REFERENCE's own code (oracle/_ref, compiled from /root/reference).
Run in the authoring container only (needs /root/reference); the outputs are committed.

  bf_reference.npz   q, t, idx1, d1 (float, as hamming32 returns), dmat — from
                     GSLAM::Vocabulary::DistanceFactory::hamming32 + the first-min scan.
  se3_reference.npz  exp/log/mul/inverse/apply samples of GSLAM::SE3 for pinning the BA pose algebra.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402


def main():
    ref = oracle_lib.load_reference()
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)

    q = oracle_lib.random_descriptors(96, 0x60D)
    base = oracle_lib.random_descriptors(128, 0x60D)  # first 96 rows equal q's stream -> correlated
    t, _ = oracle_lib.correlated_descriptors(base, 0x60E)
    t[17] = t[5]
    idx1, d1 = ref.bf_match(q, t)
    dmat = np.array([[ref.hamming32(q[i], t[j]) for j in range(16)] for i in range(16)], np.int32)
    np.savez_compressed(os.path.join(out, "bf_reference.npz"), q=q, t=t, idx1=idx1, d1=d1, dmat=dmat)

    rng = np.random.default_rng(20260923)
    xi = rng.normal(size=(64, 6)) * np.array([1, 1, 1, 0.5, 0.5, 0.5])
    xi[0, 3:] *= 1e-6   # small (non-zero) rotation
    xi[1, 3:] *= 1e-3
    poses = np.stack([ref.se3_exp(x) for x in xi])
    logs = np.stack([ref.se3_log(p) for p in poses])
    muls = np.stack([ref.se3_mul(poses[i], poses[(i + 1) % 64]) for i in range(64)])
    invs = np.stack([ref.se3_inverse(p) for p in poses])
    pts = rng.normal(size=(64, 3)) * 3
    app = np.stack([ref.se3_apply(poses[i], pts[i]) for i in range(64)])
    np.savez_compressed(os.path.join(out, "se3_reference.npz"), xi=xi, poses=poses, logs=logs, muls=muls,
                        invs=invs, pts=pts, app=app)
    print("golden vectors written to", out)


if __name__ == "__main__":
    main()
