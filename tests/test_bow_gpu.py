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


@pytest.mark.parametrize("scoring,weighting", [(0, 0), (1, 1), (2, 0), (3, 0), (4, 1), (5, 0)])
def test_bow_score_batched_vs_oracle(ctx, oracle, scoring, weighting):
    """GSLAM::Vocabulary::score for all six scoring classes (Vocabulary.h:691-979): every query against every database
    vector straight from the transform's device output; ragged counts incl. empty vectors.  Bit-identical doubles for
    L1 / L2 / chi-square / Bhattacharyya / dot; KL (logf) 1e-6 relative."""
    import torch
    from gslam_amd import bow
    voc = bow_synth.make_vocabulary(k=10, L=3, seed=3, weighting=weighting, scoring=scoring)
    v = bow.Vocabulary(ctx, voc)
    B, cap = 12, 700
    counts = np.array([700, 650, 0, 1, 699, 300, 700, 64, 65, 128, 500, 2], np.int32)
    # descriptors drawn around a shared subset of words so that the vectors overlap substantially
    desc = np.stack([bow_synth.features_near_words(voc, cap, seed=70 + (b % 4)) for b in range(B)])
    rng = np.random.default_rng(4)
    for b in range(B):
        desc[b] = desc[b][rng.permutation(cap)]
    out = v.transform(torch.from_numpy(desc).cuda(), torch.from_numpy(counts).cuda(), levelsup=1)
    bw, bv, bn = out[3], out[4], out[5]
    nq = 5
    S = bow.score(ctx, scoring, (bw[:nq], bv[:nq], bn[:nq]), (bw, bv, bn))
    torch.cuda.synchronize()
    S = S.cpu().numpy()
    hw, hv, hn = bw.cpu().numpy().view(np.uint32), bv.cpu().numpy(), bn.cpu().numpy()
    vec = [(hw[b, :hn[b]], hv[b, :hn[b]]) for b in range(B)]
    nonzero = 0
    for q in range(nq):
        for j in range(B):
            e = oracle.bow_score(scoring, vec[q], vec[j])
            if scoring == 3:
                assert (np.isnan(e) and np.isnan(S[q, j])) or abs(S[q, j] - e) <= 1e-6 * max(1.0, abs(e)), (q, j)
            else:
                assert S[q, j] == e or (np.isnan(e) and np.isnan(S[q, j])), (scoring, q, j, S[q, j], e)
            nonzero += e != 0
    assert nonzero > nq * B // 2
    # host entry point: one query against a list of host vectors
    sh = bow.score_host(ctx, scoring, vec[0], vec)
    assert np.array_equal(sh, S[0]) or scoring == 3
    v.close()


def test_bow_score_large_query_not_staged(ctx, oracle):
    """cap_q above the LDS staging limit (16384 words) takes the global-memory search path."""
    import torch
    from gslam_amd import bow
    rng = np.random.default_rng(9)
    cap = 20000
    ids_a = np.sort(rng.choice(200000, cap, replace=False)).astype(np.uint32)
    ids_b = np.sort(rng.choice(200000, 15000, replace=False)).astype(np.uint32)
    va = rng.random(cap).astype(np.float32) / cap
    vb = rng.random(15000).astype(np.float32) / 15000
    qw = torch.from_numpy(ids_a.view(np.int32)).cuda()[None]
    qv = torch.from_numpy(va).cuda()[None]
    dwn = np.full((2, cap), -1, np.int32)
    dvn = np.zeros((2, cap), np.float32)
    dwn[0, :15000] = ids_b.view(np.int32)
    dvn[0, :15000] = vb
    dwn[1] = ids_a.view(np.int32)
    dvn[1] = va
    S = bow.score(ctx, 0, (qw, qv, torch.tensor([cap], dtype=torch.int32).cuda()),
                  (torch.from_numpy(dwn).cuda(), torch.from_numpy(dvn).cuda(), torch.tensor([15000, cap], dtype=torch.int32).cuda()))
    torch.cuda.synchronize()
    S = S.cpu().numpy()
    assert S[0, 0] == oracle.bow_score(0, (ids_a, va), (ids_b, vb))
    assert S[0, 1] == oracle.bow_score(0, (ids_a, va), (ids_a, va))


@pytest.mark.parametrize("desc_bytes,k,L,weighting,scoring,levelsup,n", [(64, 10, 4, 0, 0, 2, 2000), (64, 6, 3, 1, 1, 1, 333),
                                                                         (40, 8, 3, 0, 0, 0, 1000), (8, 4, 3, 2, 5, 1, 100),
                                                                         (128, 5, 3, 3, 0, 2, 500)])
def test_bow_wide_binary_descriptors(ctx, oracle, desc_bytes, k, L, weighting, scoring, levelsup, n):
    """64-byte (hamming64) and other 8 n-byte (hamming8x) vocabularies, host entry and batched device entry: bit-identical
    to the oracle, which is pinned to the reference for these widths (tests/test_bow_oracle.py)."""
    import torch
    from gslam_amd.bow import Vocabulary
    voc = bow_synth.make_vocabulary(k=k, L=L, seed=3 + desc_bytes, weighting=weighting, scoring=scoring, desc_bytes=desc_bytes)
    rng = np.random.default_rng(desc_bytes)
    desc = np.concatenate([bow_synth.features_near_words(voc, n - n // 4, seed=5, flip_bits=min(10, desc_bytes)),
                           rng.integers(0, 256, (n // 4, desc_bytes), dtype=np.uint8)])
    v = Vocabulary(ctx, voc)
    got = v.transform_host(desc, levelsup)
    exp = oracle.bow_transform(voc, desc, levelsup)
    assert np.array_equal(got[0], exp[0]) and got[1].tobytes() == exp[1].tobytes() and np.array_equal(got[2], exp[2])
    assert np.array_equal(got[3], exp[3]) and got[4].tobytes() == exp[4].tobytes()
    out = v.transform(torch.from_numpy(np.stack([desc, desc[::-1].copy()])).cuda(), None, levelsup=levelsup)
    torch.cuda.synchronize()
    e2 = oracle.bow_transform(voc, desc[::-1].copy(), levelsup)
    assert np.array_equal(out[0][1].cpu().numpy().view(np.uint32), e2[0]) and int(out[5][1]) == len(e2[3])
    assert out[4][1, :len(e2[3])].cpu().numpy().tobytes() == e2[4].tobytes()
    v.close()


def test_bow_wide_golden_reference_vectors(ctx):
    """Directly against the reference's outputs for 64- and 40-byte descriptors (tests/golden/bow_reference_wide.npz)."""
    from gslam_amd.bow import Vocabulary
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "bow_reference_wide.npz"))
    for w in (64, 40):
        voc = bow_synth.make_vocabulary(k=int(g[f"k{w}"]), L=int(g[f"L{w}"]), seed=int(g[f"seed{w}"]), desc_bytes=w)
        v = Vocabulary(ctx, voc)
        word, weight, node, bw, bv = v.transform_host(g[f"desc{w}"], int(g[f"levelsup{w}"]))
        assert np.array_equal(word, g[f"word{w}"]) and weight.tobytes() == g[f"weight{w}"].tobytes()
        assert np.array_equal(node, g[f"node{w}"]) and np.array_equal(bw, g[f"bow_ids{w}"]) and bv.tobytes() == g[f"bow_vals{w}"].tobytes()
        v.close()


@pytest.mark.parametrize("dims,k,L,weighting,scoring,levelsup,n", [(64, 8, 3, 0, 0, 1, 2000), (128, 6, 3, 1, 1, 2, 700), (8, 5, 4, 0, 5, 0, 300)])
def test_bow_float_vocabulary(ctx, oracle, dims, k, L, weighting, scoring, levelsup, n):
    """Float (L2) vocabularies: squared L2 accumulated in float in index order without FMA, as the reference's l2generic --
    words, nodes, weights and BoW floats bit-identical to the oracle (pinned to the reference for these in
    tests/test_bow_oracle.py); then directly against the reference's golden outputs."""
    from gslam_amd.bow import Vocabulary
    voc = bow_synth.make_float_vocabulary(k=k, L=L, dims=dims, seed=dims, weighting=weighting, scoring=scoring)
    rng = np.random.default_rng(dims)
    desc = np.concatenate([bow_synth.float_features_near_words(voc, n - n // 4, seed=5), rng.normal(size=(n // 4, dims)).astype(np.float32) * 2])
    v = Vocabulary(ctx, voc)
    got = v.transform_host(desc, levelsup)
    exp = oracle.bow_transform(voc, desc, levelsup)
    assert np.array_equal(got[0], exp[0]) and got[1].tobytes() == exp[1].tobytes() and np.array_equal(got[2], exp[2])
    assert np.array_equal(got[3], exp[3]) and got[4].tobytes() == exp[4].tobytes()
    v.close()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "bow_reference_wide.npz"))
    voc = bow_synth.make_float_vocabulary(k=int(g["kf"]), L=int(g["Lf"]), dims=int(g["dimsf"]), seed=int(g["seedf"]))
    v = Vocabulary(ctx, voc)
    word, weight, node, bw, bv = v.transform_host(g["descf"], int(g["levelsupf"]))
    assert np.array_equal(word, g["wordf"]) and weight.tobytes() == g["weightf"].tobytes() and np.array_equal(node, g["nodef"])
    assert np.array_equal(bw, g["bow_idsf"]) and bv.tobytes() == g["bow_valsf"].tobytes()
    v.close()
