"""GPU parity of the BoW transform (HIP, through the C ABI) vs the oracle that is pinned to the reference's own
Vocabulary::transform: word ids, node ids, weights and the normalised BoW floats must be bit-identical."""
import os

import numpy as np
import pytest

import oracle_lib
from gslam_amd import bow_synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k,L,weighting,scoring,levelsup,n", [(10, 4, 0, 0, 2, 2000), (4, 3, 1, 1, 1, 777),
                                                              (10, 3, 2, 0, 0, 1000), (6, 4, 3, 5, 4, 300),
                                                              (10, 4, 0, 5, 7, 1500), (3, 5, 1, 2, 3, 1),
                                                              (10, 4, 0, 0, 2, 4097), (10, 4, 0, 0, 2, 8192),
                                                              (10, 4, 1, 1, 1, 16384)])
def test_bow_host_entry_parity(ctx, oracle, k, L, weighting, scoring, levelsup, n):
    from gslam_amd.bow import Vocabulary
    voc = bow_synth.make_vocabulary(k=k, L=L, seed=11 + k, weighting=weighting, scoring=scoring)
    desc = np.concatenate([bow_synth.features_near_words(voc, n - n // 4, seed=5), oracle_lib.random_descriptors(n // 4, 9)])
    v = Vocabulary(ctx, voc)
    got = v.transform_host(desc, levelsup)
    exp = oracle.bow_transform(voc, desc, levelsup)
    assert np.array_equal(got[0], exp[0]) and got[1].tobytes() == exp[1].tobytes() and np.array_equal(got[2], exp[2])
    assert np.array_equal(got[3], exp[3]) and got[4].tobytes() == exp[4].tobytes()
    v.close()


def test_bow_golden_reference_vectors(ctx):
    """Directly against the vectors produced by the reference's own code (no oracle in between)."""
    from gslam_amd.bow import Vocabulary
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "bow_reference.npz"))
    voc = bow_synth.make_vocabulary(k=int(g["k"]), L=int(g["L"]), seed=int(g["seed"]))
    v = Vocabulary(ctx, voc)
    word, weight, node, bw, bv = v.transform_host(g["desc"], int(g["levelsup"]))
    assert np.array_equal(word, g["word"]) and weight.tobytes() == g["weight"].tobytes()
    assert np.array_equal(node, g["node"]) and np.array_equal(bw, g["bow_ids"])
    assert bv.tobytes() == g["bow_vals"].tobytes()
    v.close()


def test_bow_batched_ragged_and_large_tree(ctx, oracle):
    import torch
    from gslam_amd.bow import Vocabulary
    voc = bow_synth.make_vocabulary(k=10, L=5, seed=2)  # 111 111 nodes
    v = Vocabulary(ctx, voc)
    B, cap = 5, 2000
    counts = np.array([2000, 0, 1, 1999, 640], np.int32)
    desc = np.stack([bow_synth.features_near_words(voc, cap, seed=30 + b) for b in range(B)])
    out = v.transform(torch.from_numpy(desc).cuda(), torch.from_numpy(counts).cuda(), levelsup=3)
    torch.cuda.synchronize()
    word, weight, node, bw, bv, bn = [t.cpu().numpy() for t in out]
    for b in range(B):
        n = counts[b]
        e = oracle.bow_transform(voc, desc[b, :n], 3)
        assert np.array_equal(word[b, :n].view(np.uint32), e[0]) and weight[b, :n].tobytes() == e[1].tobytes()
        assert np.array_equal(node[b, :n].view(np.uint32), e[2])
        assert bn[b] == len(e[3])
        assert np.array_equal(bw[b, :bn[b]].view(np.uint32), e[3]) and bv[b, :bn[b]].tobytes() == e[4].tobytes()
        assert (bw[b, bn[b]:].view(np.uint32) == 0xFFFFFFFF).all() and not bv[b, bn[b]:].any()
    v.close()
