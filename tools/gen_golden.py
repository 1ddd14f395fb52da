#!/usr/bin/env python3
"""Generate tests/golden/* from the REFERENCE's own code (oracle/_ref, compiled from /root/reference).
Run in the authoring container only (needs /root/reference); the outputs are committed.

  bf_reference.npz   q, t, idx1, d1 (float, as hamming32 returns), dmat — from
                     GSLAM::Vocabulary::DistanceFactory::hamming32 + the first-min scan.
  se3_reference.npz  exp/log/mul/inverse/apply samples of GSLAM::SE3 for pinning the BA pose algebra.
  sim3_reference.npz exp/log/mul/apply samples of GSLAM::SIM3 for pinning the pose-graph / alignment algebra.
  bow_reference.npz  a synthetic .gbow image (k = 10, L = 3) loaded by the reference's own Vocabulary::load; word / node
                     ids, weights, BowVector, FeatureVector of Vocabulary::transform on 500 descriptors, a second
                     BowVector and the reference's score() between the two; plus all six scoring types on that pair.
  bow_reference_wide.npz  the same for a 64-byte (hamming64) and a 40-byte (hamming8x) vocabulary (`gen_golden.py wide`).
  undist_reference.npz  remap tables of the reference's UndistorterImpl::prepareReMap (OpenCV model, 128x96 -> 112x84)
                     and its undistort / undistortFast outputs on seeded 1- and 3-channel images.
  camera_reference.npz  GSLAM::Camera::Project of a pinhole and two OpenCV cameras on seeded camera-frame points
                     (`gen_golden.py camera`): pins the projection the self-calibrating graph solve differentiates.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402


def wide_bow(ref, out):
    """bow_reference_wide.npz: 64-byte (hamming64) and 40-byte (hamming8x) vocabularies through the reference."""
    from gslam_amd import bow_synth
    rec = {}
    for w, k, L, seed, levelsup in ((64, 9, 3, 41, 1), (40, 7, 3, 42, 2)):
        voc = bow_synth.make_vocabulary(k=k, L=L, seed=seed, desc_bytes=w)
        rv = oracle_lib.RefVocabulary(ref, bow_synth.to_gbow_bytes(voc))
        rng = np.random.default_rng(seed)
        desc = np.concatenate([bow_synth.features_near_words(voc, 300, seed=seed + 1), rng.integers(0, 256, (100, w), dtype=np.uint8)])
        word, weight, node = rv.words(desc, levelsup, desc_bytes=w)
        bi, bv, fn, ff = rv.transform(desc, levelsup, desc_bytes=w)
        rv.close()
        rec.update({f"k{w}": k, f"L{w}": L, f"seed{w}": seed, f"levelsup{w}": levelsup, f"desc{w}": desc,
                    f"word{w}": word.astype(np.uint32), f"weight{w}": weight, f"node{w}": node.astype(np.uint32),
                    f"bow_ids{w}": bi.astype(np.uint32), f"bow_vals{w}": bv})
    # a float (L2) vocabulary: l2generic
    k, L, dims, seed, levelsup = 7, 3, 64, 43, 1
    voc = bow_synth.make_float_vocabulary(k=k, L=L, dims=dims, seed=seed)
    rv = oracle_lib.RefVocabulary(ref, bow_synth.to_gbow_bytes(voc))
    rng = np.random.default_rng(seed)
    desc = np.concatenate([bow_synth.float_features_near_words(voc, 300, seed=seed + 1), rng.normal(size=(100, dims)).astype(np.float32) * 2])
    bi, bv, word, weight, node = rv.transform_f32(desc, levelsup)
    rv.close()
    rec.update(dict(kf=k, Lf=L, dimsf=dims, seedf=seed, levelsupf=levelsup, descf=desc, wordf=word.astype(np.uint32), weightf=weight,
                    nodef=node.astype(np.uint32), bow_idsf=bi.astype(np.uint32), bow_valsf=bv))
    np.savez_compressed(os.path.join(out, "bow_reference_wide.npz"), **rec)
    print("bow_reference_wide.npz written")


def camera_golden(ref, out):
    """camera_reference.npz: the reference's Camera::Project (GSLAM/core/Camera.h:213-227,386-407)."""
    rng = np.random.default_rng(20260925)
    cams = np.array([[640, 480, 520.0, 515.0, 318.0, 242.0, 0, 0, 0, 0, 0],
                     [640, 480, 520.0, 515.0, 318.0, 242.0, -0.28, 0.09, 1.2e-3, -8e-4, -0.01],
                     [1920, 1080, 1400.0, 1395.0, 955.0, 545.0, 0.12, -0.3, -2e-3, 1.5e-3, 0.2]])
    xyz = np.stack([rng.uniform(-3, 3, 400), rng.uniform(-3, 3, 400), rng.uniform(2, 9, 400)], axis=1)
    xyz[:40, 2] = 1.0  # (the reference short-cuts z == 1)
    uv = np.zeros((len(cams), len(xyz), 2))
    for i, c in enumerate(cams):
        params = c[:6] if not c[6:].any() else c  # 6 parameters = CameraPinhole, 11 = CameraOpenCV
        assert ref.camera_project(params, xyz, uv[i])
    np.savez_compressed(os.path.join(out, "camera_reference.npz"), cams=cams, xyz=xyz, uv=uv)
    print("camera_reference.npz written")


def bf_bytes_golden(ref, out):
    """bf_bytes_reference.npz: the reference's hamming64 / hamming8x (Vocabulary.h:493-513) + its first-minimum scan on rows of
    64, 40, 16 and 128 bytes (with duplicated train rows: the tie rule)"""
    rng = np.random.default_rng(0xB17E5)
    data = {}
    for nb in (64, 40, 16, 128):
        q = rng.integers(0, 256, size=(48, nb), dtype=np.uint8)
        t = rng.integers(0, 256, size=(61, nb), dtype=np.uint8)
        t[:40] = q[:40] ^ (rng.random((40, nb)) < 0.04).astype(np.uint8) * np.uint8(1 << 3)  # correlated rows
        t[50] = t[9]
        idx1, d1 = ref.bf_match_bytes(q, t, nb)
        data.update({"q%d" % nb: q, "t%d" % nb: t, "idx%d" % nb: idx1, "d%d" % nb: d1})
    np.savez_compressed(os.path.join(out, "bf_bytes_reference.npz"), **data)
    print("bf_bytes_reference.npz written")


def main():
    ref = oracle_lib.load_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "bf_bytes":
        return bf_bytes_golden(ref, os.path.join(ROOT, "tests", "golden"))
    if len(sys.argv) > 1 and sys.argv[1] == "wide":
        return wide_bow(ref, os.path.join(ROOT, "tests", "golden"))
    if len(sys.argv) > 1 and sys.argv[1] == "camera":
        return camera_golden(ref, os.path.join(ROOT, "tests", "golden"))
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
    # ---- SIM3 algebra of the pose-graph / alignment oracle: the reference's own exp / log / operator* / apply
    mu = rng.normal(size=(64, 7)) * np.array([3, 3, 3, 0.6, 0.6, 0.6, 0.5])
    mu[0, 3:6] *= 1e-4   # small rotation
    mu[1, 6] = 0.02      # small (not tiny) log-scale; the reference loses digits below ~1e-6 ((s - 1) / sigma)
    mu[2, 3:6] *= 4.0    # large rotation
    sims = np.stack([ref.sim3_exp(m) for m in mu])
    slogs = np.stack([ref.sim3_log(q) for q in sims])
    smuls = np.stack([ref.sim3_mul(sims[i], sims[(i + 1) % 64]) for i in range(64)])
    sapp = np.stack([ref.sim3_apply(sims[i], pts[i]) for i in range(64)])
    np.savez_compressed(os.path.join(out, "sim3_reference.npz"), mu=mu, sims=sims, logs=slogs, muls=smuls, pts=pts, app=sapp)
    # ---- BoW: the reference's own Vocabulary::load / transform / score on a synthetic vocabulary image
    from gslam_amd import bow_synth
    k, L, seed, levelsup = 10, 3, 7, 1
    voc = bow_synth.make_vocabulary(k=k, L=L, seed=seed)
    gbow = bow_synth.to_gbow_bytes(voc)
    rv = oracle_lib.RefVocabulary(ref, gbow)
    assert rv.info() == (k, L, len(voc["nodes"]))
    desc = np.concatenate([bow_synth.features_near_words(voc, 400, seed=41), oracle_lib.random_descriptors(100, 0xB0)])
    desc2 = np.concatenate([bow_synth.features_near_words(voc, 330, seed=42), oracle_lib.random_descriptors(70, 0xB1)])
    word, weight, node = rv.words(desc, levelsup)
    bi, bv, fn, ff = rv.transform(desc, levelsup)
    bi2, bv2, _, _ = rv.transform(desc2, levelsup)
    score12 = rv.score((bi, bv), (bi2, bv2))
    rv.close()
    # the same pair under every scoring type of the reference (the vocabulary's scoring object is chosen at load)
    scores = np.zeros(6)
    for sc in range(6):
        v2 = dict(voc, scoring=sc)
        r2 = oracle_lib.RefVocabulary(ref, bow_synth.to_gbow_bytes(v2))
        a = r2.transform(desc, levelsup)
        b = r2.transform(desc2, levelsup)
        scores[sc] = r2.score((a[0], a[1]), (b[0], b[1]))
        r2.close()
    np.savez_compressed(os.path.join(out, "bow_reference.npz"), k=k, L=L, seed=seed, levelsup=levelsup,
                        gbow=np.frombuffer(gbow, np.uint8), desc=desc, desc2=desc2, word=word.astype(np.uint32),
                        weight=weight, node=node.astype(np.uint32), bow_ids=bi.astype(np.uint32), bow_vals=bv,
                        fv_nodes=fn, fv_feat=ff, bow2_ids=bi2.astype(np.uint32), bow2_vals=bv2, score12=score12,
                        scores_by_type=scores)

    # ---- undistortion: tables and outputs of the reference's own UndistorterImpl
    cam_in = [128, 96, 100, 101, 63.5, 47.2, -0.31, 0.11, 0.001, -0.0007, -0.02]
    cam_out = [112, 84, 80, 80, 56, 42]
    ru = oracle_lib.RefUndistorter(ref, cam_in, cam_out)
    t = ru.tables()
    rng = np.random.default_rng(20260924)
    img1 = rng.integers(0, 256, (ru.h_in, ru.w_in), dtype=np.uint8)
    img3 = rng.integers(0, 256, (ru.h_in, ru.w_in, 3), dtype=np.uint8)
    rec = dict(w_in=ru.w_in, h_in=ru.h_in, w_out=ru.w_out, h_out=ru.h_out, cam_in=np.array(cam_in), cam_out=np.array(cam_out),
               remapX=t["remapX"], remapFast=t["remapFast"], remapIdx=t["remapIdx"], remapCoef=t["remapCoef"],
               img1=img1, img3=img3)
    for ch, img in ((1, img1), (3, img3)):
        for fast in (0, 1):
            rec[f"out{ch}_{fast}"] = ru.run(img, fast=bool(fast))
    ru.close()
    np.savez_compressed(os.path.join(out, "undist_reference.npz"), **rec)
    wide_bow(ref, out)
    camera_golden(ref, out)
    print("golden vectors written to", out)


if __name__ == "__main__":
    main()
