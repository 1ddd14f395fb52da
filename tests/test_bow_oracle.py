"""BoW transform oracle PINNED to the reference: GSLAM::Vocabulary::load + transform compiled from
/root/reference (oracle/_ref) live when available, and the committed golden vectors generated from it."""
import os

import numpy as np
import pytest

import oracle_lib
from gslam_amd import bow_synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "bow_reference.npz")


def _fv_pairs(node, weight):
    """FeatureVector in map order: (node id asc, feature index asc) for features with weight > 0."""
    idx = np.nonzero(weight > 0)[0]
    order = np.lexsort((idx, node[idx]))
    return node[idx][order].astype(np.uint64), idx[order].astype(np.uint32)


def test_golden_vectors_from_reference(oracle):
    g = np.load(GOLD)
    voc = bow_synth.make_vocabulary(k=int(g["k"]), L=int(g["L"]), seed=int(g["seed"]))
    assert bow_synth.to_gbow_bytes(voc) == g["gbow"].tobytes()  # the generator is reproducible
    word, weight, node, bw, bv = oracle.bow_transform(voc, g["desc"], levelsup=int(g["levelsup"]))
    assert np.array_equal(word, g["word"]) and np.array_equal(weight, g["weight"]) and np.array_equal(node, g["node"])
    assert np.array_equal(bw, g["bow_ids"]) and bv.tobytes() == g["bow_vals"].tobytes()  # bit-exact floats
    fn, ff = _fv_pairs(node, weight)
    assert np.array_equal(fn, g["fv_nodes"]) and np.array_equal(ff, g["fv_feat"])
    assert abs(oracle.bow_score_l1((bw, bv), (g["bow2_ids"], g["bow2_vals"])) - float(g["score12"])) == 0.0
    # all six scoring classes of the reference on the same descriptor pair (each vocabulary normalises its own way)
    for sc in range(6):
        v2 = dict(voc, scoring=sc)
        a = oracle.bow_transform(v2, g["desc"], levelsup=int(g["levelsup"]))
        b = oracle.bow_transform(v2, g["desc2"], levelsup=int(g["levelsup"]))
        assert oracle.bow_score(sc, (a[3], a[4]), (b[3], b[4])) == float(g["scores_by_type"][sc]), sc


@pytest.mark.skipif(not oracle_lib.have_reference(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("k,L,weighting,scoring,levelsup", [(10, 4, 0, 0, 2), (4, 3, 1, 1, 1), (10, 3, 2, 0, 0),
                                                            (6, 4, 3, 5, 4), (10, 4, 0, 5, 7), (3, 5, 1, 2, 3),
                                                            (8, 3, 0, 3, 1), (8, 3, 1, 4, 2)])
def test_oracle_equals_reference_live(oracle, k, L, weighting, scoring, levelsup):
    ref = oracle_lib.load_reference()
    voc = bow_synth.make_vocabulary(k=k, L=L, seed=11 + k, weighting=weighting, scoring=scoring)
    rv = oracle_lib.RefVocabulary(ref, bow_synth.to_gbow_bytes(voc))
    assert rv.info() == (k, L, len(voc["nodes"]))
    desc = np.concatenate([bow_synth.features_near_words(voc, 700, seed=5), oracle_lib.random_descriptors(300, 9)])
    word, weight, node, bw, bv = oracle.bow_transform(voc, desc, levelsup=levelsup)
    rw, rwt, rn = rv.words(desc, levelsup)
    assert np.array_equal(word, rw) and np.array_equal(weight, rwt) and np.array_equal(node, rn)
    bi, bvr, fn, ff = rv.transform(desc, levelsup)
    assert np.array_equal(bw, bi) and bv.tobytes() == bvr.tobytes()
    en, ef = _fv_pairs(node, weight)
    assert np.array_equal(en, fn) and np.array_equal(ef, ff)
    desc2 = bow_synth.features_near_words(voc, 800, seed=6)
    _, _, _, bw2, bv2 = oracle.bow_transform(voc, desc2, levelsup=levelsup)
    if scoring == 0:
        assert oracle.bow_score_l1((bw, bv), (bw2, bv2)) == rv.score((bi, bvr), (bw2.astype(np.uint64), bv2))
    # the vocabulary's own scoring object (chosen by `scoring` at load), both argument orders (KL is asymmetric)
    assert oracle.bow_score(scoring, (bw, bv), (bw2, bv2)) == rv.score((bi, bvr), (bw2.astype(np.uint64), bv2))
    assert oracle.bow_score(scoring, (bw2, bv2), (bw, bv)) == rv.score((bw2.astype(np.uint64), bv2), (bi, bvr))
    assert oracle.bow_score(scoring, (bw, bv), (bw, bv)) == rv.score((bi, bvr), (bi, bvr))
    rv.close()


def test_stopped_words_and_empty_input(oracle):
    voc = bow_synth.make_vocabulary(k=4, L=2, seed=3, stop_frac=0.5)
    desc = bow_synth.features_near_words(voc, 200, seed=1)
    word, weight, node, bw, bv = oracle.bow_transform(voc, desc)
    assert (weight == 0).any() and not np.isin(word[weight == 0], bw).any()
    assert abs(np.abs(bv).sum() - 1.0) < 1e-5
    w0 = oracle.bow_transform(voc, np.zeros((0, 32), np.uint8))
    assert len(w0[3]) == 0


@pytest.mark.skipif(not oracle_lib.have_reference(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("desc_bytes,k,L,weighting,scoring,levelsup", [(64, 10, 3, 0, 0, 1), (64, 6, 4, 1, 1, 2), (40, 8, 3, 0, 0, 0),
                                                                       (8, 4, 3, 2, 5, 1), (128, 5, 3, 3, 0, 2)])
def test_wide_binary_descriptors_equal_the_reference_live(oracle, desc_bytes, k, L, weighting, scoring, levelsup):
    """64-byte descriptors take the reference's hamming64, other multiples of 8 its hamming8x (Vocabulary.h:493-513,560-568)."""
    ref = oracle_lib.load_reference()
    voc = bow_synth.make_vocabulary(k=k, L=L, seed=3 + desc_bytes, weighting=weighting, scoring=scoring, desc_bytes=desc_bytes)
    rv = oracle_lib.RefVocabulary(ref, bow_synth.to_gbow_bytes(voc))
    assert rv.info() == (k, L, len(voc["nodes"]))
    rng = np.random.default_rng(desc_bytes)
    desc = np.concatenate([bow_synth.features_near_words(voc, 500, seed=5, flip_bits=min(10, desc_bytes)),
                           rng.integers(0, 256, (200, desc_bytes), dtype=np.uint8)])
    word, weight, node, bw, bv = oracle.bow_transform(voc, desc, levelsup=levelsup)
    rw, rwt, rn = rv.words(desc, levelsup, desc_bytes=desc_bytes)
    assert np.array_equal(word, rw) and np.array_equal(weight, rwt) and np.array_equal(node, rn)
    bi, bvr, fn, ff = rv.transform(desc, levelsup, desc_bytes=desc_bytes)
    assert np.array_equal(bw, bi) and bv.tobytes() == bvr.tobytes()
    en, ef = _fv_pairs(node, weight)
    assert np.array_equal(en, fn) and np.array_equal(ef, ff)
    assert len(np.unique(word)) > 20
    rv.close()


def test_wide_descriptor_golden_vectors(oracle):
    """The same pinned on fixtures that travel to the GPU box (tools/gen_golden.py, from oracle/_ref)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "bow_reference_wide.npz"))
    for w in (64, 40):
        voc = bow_synth.make_vocabulary(k=int(g[f"k{w}"]), L=int(g[f"L{w}"]), seed=int(g[f"seed{w}"]), desc_bytes=w)
        word, weight, node, bw, bv = oracle.bow_transform(voc, g[f"desc{w}"], levelsup=int(g[f"levelsup{w}"]))
        assert np.array_equal(word, g[f"word{w}"]) and np.array_equal(node, g[f"node{w}"])
        assert np.array_equal(bw, g[f"bow_ids{w}"]) and bv.tobytes() == g[f"bow_vals{w}"].tobytes()


@pytest.mark.skipif(not oracle_lib.have_reference(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("dims,k,L,weighting,scoring,levelsup", [(64, 8, 3, 0, 0, 1), (128, 6, 3, 1, 1, 2), (8, 5, 4, 0, 5, 0), (24, 4, 3, 2, 0, 1)])
def test_float_vocabulary_equals_the_reference_live(oracle, dims, k, L, weighting, scoring, levelsup):
    """Float (L2) vocabularies: the reference's l2generic (Vocabulary.h:550-560) through oracle/_ref, bit for bit."""
    ref = oracle_lib.load_reference()
    voc = bow_synth.make_float_vocabulary(k=k, L=L, dims=dims, seed=dims, weighting=weighting, scoring=scoring)
    rv = oracle_lib.RefVocabulary(ref, bow_synth.to_gbow_bytes(voc))
    assert rv.info() == (k, L, len(voc["nodes"]))
    rng = np.random.default_rng(dims)
    desc = np.concatenate([bow_synth.float_features_near_words(voc, 500, seed=5), rng.normal(size=(200, dims)).astype(np.float32) * 2])
    word, weight, node, bw, bv = oracle.bow_transform(voc, desc, levelsup=levelsup)
    bi, bvr, rw, rwt, rn = rv.transform_f32(desc, levelsup)
    assert np.array_equal(word, rw) and np.array_equal(weight, rwt) and np.array_equal(node, rn)
    assert np.array_equal(bw, bi) and bv.tobytes() == bvr.tobytes()
    assert len(np.unique(word)) > 20
    rv.close()


def test_float_vocabulary_golden_vectors(oracle):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "bow_reference_wide.npz"))
    voc = bow_synth.make_float_vocabulary(k=int(g["kf"]), L=int(g["Lf"]), dims=int(g["dimsf"]), seed=int(g["seedf"]))
    word, weight, node, bw, bv = oracle.bow_transform(voc, g["descf"], levelsup=int(g["levelsupf"]))
    assert np.array_equal(word, g["wordf"]) and np.array_equal(node, g["nodef"])
    assert np.array_equal(bw, g["bow_idsf"]) and bv.tobytes() == g["bow_valsf"].tobytes()
