"""GPU parity: gh_bf_* (HIP, through the C ABI) vs the CPU oracle — bit-exact indices and distances."""
import os

import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu


def _u16(t):
    return t.cpu().numpy().view(np.uint16)


def _run(ctx, q, t):
    import torch
    from gslam_amd.matcher import BFMatcher
    m = BFMatcher(ctx)
    dq = torch.from_numpy(q).cuda()
    dt = torch.from_numpy(t).cuda() if t.shape[0] else torch.zeros((0, 32), dtype=torch.uint8, device="cuda")
    idx1, d1, d2 = m.match(dq, dt)
    torch.cuda.synchronize()
    return idx1.cpu().numpy(), _u16(d1), _u16(d2)


@pytest.mark.parametrize("nq,nt", [(1, 1), (1, 5), (63, 64), (64, 3), (129, 130), (257, 301), (2000, 2000),
                                   (1000, 4099)])
def test_bf_parity_random(ctx, oracle, nq, nt):
    q = oracle_lib.random_descriptors(nq, 100 + nq)
    t = oracle_lib.random_descriptors(nt, 200 + nt)
    got = _run(ctx, q, t)
    exp = oracle.bf_match(q, t, threads=4)
    for g, e in zip(got, exp):
        assert np.array_equal(g, e)


def test_bf_parity_correlated_and_ties(ctx, oracle):
    base = oracle_lib.random_descriptors(2000, 7)
    t, perm = oracle_lib.correlated_descriptors(base, 8)
    t[100] = t[50]
    t[1999] = t[0]
    got = _run(ctx, base, t)
    exp = oracle.bf_match(base, t, threads=4)
    for g, e in zip(got, exp):
        assert np.array_equal(g, e)
    # duplicate rows: the lower index must win
    q = t[[100, 1999]]
    idx1, d1, d2 = _run(ctx, q, t)
    assert idx1.tolist() == [50, 0] and d1.tolist() == [0, 0] and d2.tolist() == [0, 0]


def test_bf_golden_reference_vectors(ctx):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "bf_reference.npz"))
    idx1, d1, d2 = _run(ctx, g["q"], g["t"])
    assert np.array_equal(idx1, g["idx1"])
    assert np.array_equal(d1.astype(np.float32), g["d1"])


def test_bf_empty_train_and_single(ctx):
    q = oracle_lib.random_descriptors(5, 1)
    idx1, d1, d2 = _run(ctx, q, np.zeros((0, 32), np.uint8))
    assert (idx1 == -1).all() and (d1 == 65535).all() and (d2 == 65535).all()
    idx1, d1, d2 = _run(ctx, q, q[:1])
    assert (idx1 == 0).all() and d1[0] == 0 and (d2 == 65535).all()


def test_bf_host_entry_point(ctx, oracle):
    import ctypes as C
    from gslam_amd import hip
    q = oracle_lib.random_descriptors(300, 31)
    t = oracle_lib.random_descriptors(500, 32)
    idx1 = np.empty(300, np.int32)
    d1 = np.empty(300, np.uint16)
    d2 = np.empty(300, np.uint16)
    ctx.check(hip.lib.gh_bf_match_host(ctx.h, q.ctypes.data_as(C.c_void_p), 300, t.ctypes.data_as(C.c_void_p), 500,
                                       idx1.ctypes.data_as(C.c_void_p), d1.ctypes.data_as(C.c_void_p),
                                       d2.ctypes.data_as(C.c_void_p)))
    e = oracle.bf_match(q, t)
    assert np.array_equal(idx1, e[0]) and np.array_equal(d1, e[1]) and np.array_equal(d2, e[2])


def test_bf_pairs_ragged_counts(ctx, oracle):
    import torch
    from gslam_amd.matcher import BFMatcher
    F, cap = 6, 300
    counts = np.array([300, 0, 1, 129, 257, 64], np.int32)
    desc = np.stack([oracle_lib.random_descriptors(cap, 900 + f) for f in range(F)])
    pq = np.array([0, 1, 2, 3, 4, 5, 0, 3], np.int32)
    pt = np.array([3, 0, 0, 4, 5, 1, 0, 2], np.int32)
    m = BFMatcher(ctx)
    idx1, d1, d2 = m.match_pairs(torch.from_numpy(desc).cuda(), torch.from_numpy(counts).cuda(),
                                 torch.from_numpy(pq).cuda(), torch.from_numpy(pt).cuda())
    torch.cuda.synchronize()
    idx1, d1, d2 = idx1.cpu().numpy(), _u16(d1), _u16(d2)
    for p in range(len(pq)):
        nq, nt = counts[pq[p]], counts[pt[p]]
        e = oracle.bf_match(desc[pq[p], :nq], desc[pt[p], :nt])
        assert np.array_equal(idx1[p, :nq], e[0])
        assert np.array_equal(d1[p, :nq], e[1])
        assert np.array_equal(d2[p, :nq], e[2])
        assert (idx1[p, nq:] == -1).all() and (d1[p, nq:] == 65535).all() and (d2[p, nq:] == 65535).all()


def test_match_mask_parity(ctx, oracle):
    import torch
    from gslam_amd.matcher import BFMatcher
    base = oracle_lib.random_descriptors(1500, 17)
    t, _ = oracle_lib.correlated_descriptors(base, 18)
    m = BFMatcher(ctx)
    dq, dt = torch.from_numpy(base).cuda(), torch.from_numpy(t).cuda()
    f = m.match(dq, dt)
    b = m.match(dt, dq)
    keep = m.mask(f[0], f[1], f[2], back_idx1=b[0], nt=1500, max_dist=80, ratio_num=8, ratio_den=10,
                  cross_check=True)
    torch.cuda.synchronize()
    fo = oracle.bf_match(base, t, threads=4)
    bo = oracle.bf_match(t, base, threads=4)
    ko = oracle.match_mask(fo[0], fo[1], fo[2], bo[0], 1500, 80, 8, 10, 1)
    assert np.array_equal(keep.cpu().numpy(), ko)
    assert 0 < ko.sum() < 1500


def test_bf_full_size_property(ctx):
    """C2 size (2000 x 2000 per pair, many pairs): size-independent properties instead of the oracle:
    matching a set against a row-permutation of itself must return the inverse permutation with d1 = 0."""
    import torch
    from gslam_amd.matcher import BFMatcher
    F, cap = 8, 2000
    g = torch.Generator(device="cpu").manual_seed(5)
    desc = torch.randint(0, 256, (F, cap, 32), dtype=torch.uint8, generator=g)
    perm = torch.stack([torch.randperm(cap, generator=g) for _ in range(F)])
    shuf = torch.stack([desc[f][perm[f]] for f in range(F)])
    allf = torch.cat([desc, shuf]).cuda()
    counts = torch.full((2 * F,), cap, dtype=torch.int32).cuda()
    pq = torch.arange(F, dtype=torch.int32).cuda()
    pt = (torch.arange(F, dtype=torch.int32) + F).cuda()
    idx1, d1, d2 = BFMatcher(ctx).match_pairs(allf, counts, pq, pt)
    torch.cuda.synchronize()
    inv = torch.argsort(perm, dim=1).to(torch.int32)
    assert torch.equal(idx1.cpu(), inv)
    assert int(d1.cpu().abs().sum()) == 0
    assert (d2.cpu().to(torch.int32) > 60).all()


def test_mfma_formulation_is_bit_identical_to_the_popcount_kernel(ctx, oracle):
    """gh_bf_match_pairs_mfma_dev (hamming = |a| + |b| - 2 |a & b| on v_mfma_i32_16x16x64_i8) against the popcount kernel
    and the oracle: idx1, d1, d2 for every row, incl. ragged counts (0, 1, 2, 17, cap), correlated descriptors with forced
    ties (lowest index must win), all-zero / all-one descriptors and capacities that are not multiples of 16."""
    import torch
    from gslam_amd.matcher import BFMatcher
    m = BFMatcher(ctx)
    for cap, counts in ((300, [300, 0, 1, 2, 17, 299, 300, 64]), (2000, [2000, 1999, 1234, 2000]), (37, [37, 5, 16, 33])):
        F = len(counts)
        desc = np.zeros((F, cap, 32), np.uint8)
        base = oracle_lib.random_descriptors(cap, 0xBEEF + cap)
        for f, n in enumerate(counts):
            d, _ = oracle_lib.correlated_descriptors(base, 77 + f)
            desc[f, :n] = d[:n]
        desc[0, 3] = desc[0, 1]      # exact duplicates: the first must win
        desc[0, 5] = 0
        desc[0, 6] = 255
        if cap > 40:
            desc[3, 40] = desc[0, 10]
            desc[3, 7] = desc[0, 10]
        pq = np.array([(i, j) for i in range(F) for j in range(F)], np.int32)
        dd = torch.from_numpy(desc).cuda()
        cc = torch.tensor(counts, dtype=torch.int32, device="cuda")
        q, t = torch.from_numpy(pq[:, 0].copy()).cuda(), torch.from_numpy(pq[:, 1].copy()).cuda()
        a = m.match_pairs(dd, cc, q, t, mfma=False)
        b = m.match_pairs(dd, cc, q, t, mfma=True)
        torch.cuda.synchronize()
        for x, y, name in zip(a, b, ("idx1", "d1", "d2")):
            assert torch.equal(x, y), (cap, name, (x != y).nonzero()[:5].tolist())
        # and against the oracle for a few pairs
        for p in (0, 1, F + 2, len(pq) - 1):
            i, j = pq[p]
            e = oracle.bf_match(desc[i, :counts[i]], desc[j, :counts[j]])
            n = counts[i]
            assert np.array_equal(b[0][p, :n].cpu().numpy(), e[0]) and np.array_equal(b[1][p, :n].cpu().numpy().view(np.uint16), e[1])
            assert np.array_equal(b[2][p, :n].cpu().numpy().view(np.uint16), e[2])


def test_mfma_persistent_workgroups_walk_many_pairs(ctx):
    """More frame pairs than workgroup columns (the workgroups loop over pairs and reuse their LDS stages, with inactive
    waves and empty frames in between): every row equals the popcount kernel's."""
    import torch
    from gslam_amd.matcher import BFMatcher
    m = BFMatcher(ctx)
    F, cap = 48, 150
    g = torch.Generator(device="cuda").manual_seed(11)
    desc = torch.randint(0, 256, (F, cap, 32), dtype=torch.uint8, device="cuda", generator=g)
    desc[:, 1::2] &= desc[:, 0::2][:, : desc[:, 1::2].shape[1]]  # correlated rows: small distances and ties
    counts = torch.randint(0, cap + 1, (F,), dtype=torch.int32, device="cuda", generator=g)
    counts[3] = 0
    counts[4] = cap
    q = torch.arange(F, dtype=torch.int32, device="cuda").repeat_interleave(F)
    t = torch.arange(F, dtype=torch.int32, device="cuda").repeat(F)
    assert q.shape[0] > 4 * 256  # more pairs than persistent workgroup columns
    a = m.match_pairs(desc, counts, q, t, mfma=False)
    b = m.match_pairs(desc, counts, q, t, mfma=True)
    torch.cuda.synchronize()
    for x, y, name in zip(a, b, ("idx1", "d1", "d2")):
        assert torch.equal(x, y), (name, (x != y).nonzero()[:5].tolist())


def test_mfma_matcher_fuzz_against_the_popcount_kernel(ctx):
    """Random capacities (incl. non-multiples of 16 / 64 / 512), ragged counts with zeros and ones, correlated rows, random
    pair lists with repeats and self-pairs: the MFMA formulation must return the popcount kernel's rows bit for bit."""
    import torch
    from hypothesis import given, settings, strategies as st
    from gslam_amd.matcher import BFMatcher
    m = BFMatcher(ctx)

    @settings(max_examples=60, deadline=None)
    @given(cap=st.integers(1, 700), frames=st.integers(1, 6), npairs=st.integers(1, 12), seed=st.integers(0, 2 ** 31 - 1),
           corr=st.booleans())
    def run(cap, frames, npairs, seed, corr):
        g = torch.Generator(device="cuda").manual_seed(seed)
        desc = torch.randint(0, 256, (frames, cap, 32), dtype=torch.uint8, device="cuda", generator=g)
        if corr and cap > 1:  # few distinct rows: many exact ties
            desc = desc[:, torch.randint(0, max(1, cap // 8), (cap,), device="cuda", generator=g)]
            desc = desc.contiguous()
        counts = torch.randint(0, cap + 1, (frames,), dtype=torch.int32, device="cuda", generator=g)
        counts[torch.randint(0, frames, (1,), device="cuda", generator=g)] = cap
        q = torch.randint(0, frames, (npairs,), dtype=torch.int32, device="cuda", generator=g)
        t = torch.randint(0, frames, (npairs,), dtype=torch.int32, device="cuda", generator=g)
        a = m.match_pairs(desc, counts, q, t, mfma=False)
        b = m.match_pairs(desc, counts, q, t, mfma=True)
        torch.cuda.synchronize()
        for x, y, name in zip(a, b, ("idx1", "d1", "d2")):
            assert torch.equal(x, y), (cap, frames, npairs, seed, name, (x != y).nonzero()[:4].tolist())

    run()


def test_pairs_entry_dispatches_on_pair_work_and_both_routes_agree(ctx):
    """gh_bf_match_pairs_dev (what the plugins and bench.py call): a small batch runs the popcount kernel, a batch with
    enough pair work the MFMA kernel (seen in the per-kernel profile); either way the rows equal the popcount entry's."""
    import torch
    from gslam_amd.matcher import BFMatcher
    m = BFMatcher(ctx)
    g = torch.Generator(device="cuda").manual_seed(5)
    for F, cap, want in ((4, 300, "bf_match_pairs"), (40, 2000, "bf_match_pairs_mfma")):
        desc = torch.randint(0, 256, (F, cap, 32), dtype=torch.uint8, device="cuda", generator=g)
        desc[:, 1::2] &= desc[:, 0::2]
        counts = torch.randint(cap // 2, cap + 1, (F,), dtype=torch.int32, device="cuda", generator=g)
        q = torch.arange(F - 1, dtype=torch.int32, device="cuda")
        t = q + 1
        ctx.prof_enable(True)
        a = m.match_pairs(desc, counts, q, t)
        used = ctx.prof_collect()
        ctx.prof_enable(False)
        assert want in used and len([k for k in used if k.startswith("bf_match")]) == 1, used.keys()
        b = m.match_pairs(desc, counts, q, t, mfma=False)
        torch.cuda.synchronize()
        for x, y, name in zip(a, b, ("idx1", "d1", "d2")):
            assert torch.equal(x, y), (F, cap, name)


def test_train_set_beyond_65535_rows(ctx, oracle):
    """A frame against a local map: nt = 150 000 (three chunks of the 16-bit index field).  Exact duplicates of the best row
    planted in later chunks must not displace the first one and must show up as the second-best distance; a strictly better
    row in the last chunk must win.  Device and host entries, vs the oracle."""
    import torch
    from gslam_amd import hip
    from gslam_amd.matcher import BFMatcher
    import ctypes as C
    rng = np.random.default_rng(7)
    nq, nt = 300, 150000
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    t[1000] = q[0]; t[70000] = q[0]; t[140000] = q[0]        # three exact copies: index 1000 wins, d2 = 0
    t[66000] = q[1]; t[66000, 0] ^= 1; t[149999] = q[1]       # distance 1 in chunk 1, distance 0 in the last row: the last row wins, d2 = 1
    t[65531] = q[2]; t[65532] = q[2]                          # duplicates across the chunk boundary (65532 rows per chunk)
    e = oracle.bf_match(q, t, threads=8)
    assert e[0][0] == 1000 and e[2][0] == 0 and e[0][1] == 149999 and e[2][1] == 1 and e[0][2] == 65531
    m = BFMatcher(ctx)
    idx1, d1, d2 = m.match(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(idx1.cpu().numpy(), e[0])
    assert np.array_equal(d1.cpu().numpy().view(np.uint16), e[1]) and np.array_equal(d2.cpu().numpy().view(np.uint16), e[2])
    hi, h1, h2 = np.empty(nq, np.int32), np.empty(nq, np.uint16), np.empty(nq, np.uint16)
    ctx.check(hip.lib.gh_bf_match_host(ctx.h, q.ctypes.data_as(C.c_void_p), nq, t.ctypes.data_as(C.c_void_p), nt,
                                       hi.ctypes.data_as(C.c_void_p), h1.ctypes.data_as(C.c_void_p), h2.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(hi, e[0]) and np.array_equal(h1, e[1]) and np.array_equal(h2, e[2])


# ------------------------------------------------------------------ wider descriptors (hamming64 / hamming8x)
@pytest.mark.parametrize("nb,nq,nt", [(64, 700, 900), (64, 1, 1), (16, 130, 257), (40, 333, 100), (128, 200, 300), (256, 65, 64),
                                      (8, 64, 1000), (64, 5, 0)])
def test_wide_descriptors_match_the_oracle(ctx, oracle, nb, nq, nt):
    """gh_bf_match_bytes_dev (GSLAM/core/Vocabulary.h:493-513: hamming64 for 64-byte rows, hamming8x for other multiples of 8)
    against the oracle, which is pinned to the reference's own functions: indices, best and second-best distances bit for bit,
    duplicated train rows (first minimum) included."""
    import torch
    from gslam_amd.matcher import BFMatcher
    rng = np.random.default_rng(nb * 1000 + nq)
    q = rng.integers(0, 256, size=(nq, nb), dtype=np.uint8)
    t = rng.integers(0, 256, size=(nt, nb), dtype=np.uint8)
    if nt > 40 and nq > 20:
        t[:20] = q[:20] ^ (rng.random((20, nb)) < 0.05).astype(np.uint8)
        t[33] = t[4]
    m = BFMatcher(ctx)
    idx1, d1, d2 = m.match_bytes(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda())
    torch.cuda.synchronize()
    e = oracle.bf_match_bytes(q, t, nb)
    assert np.array_equal(idx1.cpu().numpy(), e[0])
    assert np.array_equal(d1.cpu().numpy().view(np.uint16), e[1]) and np.array_equal(d2.cpu().numpy().view(np.uint16), e[2])


def test_wide_descriptor_batched_pairs(ctx, oracle):
    import torch
    from gslam_amd.matcher import BFMatcher
    rng = np.random.default_rng(9)
    F, cap, nb = 5, 300, 64
    desc = rng.integers(0, 256, size=(F, cap, nb), dtype=np.uint8)
    counts = np.array([300, 120, 0, 299, 64], np.int32)
    pq = np.array([0, 1, 3, 4, 2, 0], np.int32)
    pt = np.array([1, 0, 4, 3, 0, 2], np.int32)
    m = BFMatcher(ctx)
    idx1, d1, d2 = m.match_pairs_bytes(torch.from_numpy(desc).cuda(), torch.from_numpy(counts).cuda(), torch.from_numpy(pq).cuda(),
                                       torch.from_numpy(pt).cuda())
    torch.cuda.synchronize()
    for p in range(len(pq)):
        nq_, nt_ = counts[pq[p]], counts[pt[p]]
        e = oracle.bf_match_bytes(desc[pq[p], :nq_], desc[pt[p], :nt_], nb)
        assert np.array_equal(idx1[p, :nq_].cpu().numpy(), e[0]) and bool((idx1[p, nq_:] == -1).all())
        assert np.array_equal(d1[p, :nq_].cpu().numpy().view(np.uint16), e[1])
        assert np.array_equal(d2[p, :nq_].cpu().numpy().view(np.uint16), e[2])
    # 32-byte rows through the generic entries are the kernels of the default path
    d32 = rng.integers(0, 256, size=(2, 200, 32), dtype=np.uint8)
    c32 = np.array([200, 150], np.int32)
    a = m.match_pairs_bytes(torch.from_numpy(d32).cuda(), torch.from_numpy(c32).cuda(), torch.tensor([0], dtype=torch.int32).cuda(),
                            torch.tensor([1], dtype=torch.int32).cuda())
    b = m.match_pairs(torch.from_numpy(d32).cuda(), torch.from_numpy(c32).cuda(), torch.tensor([0], dtype=torch.int32).cuda(),
                      torch.tensor([1], dtype=torch.int32).cuda())
    assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_wide_descriptors_host_entry(ctx, oracle):
    """gh_bf_match_bytes_host: what FeatureDetector::match of the plugin calls for descriptors that are not 32 bytes wide."""
    import ctypes as C
    from gslam_amd import hip
    rng = np.random.default_rng(77)
    for nb in (64, 24):
        q = rng.integers(0, 256, size=(300, nb), dtype=np.uint8)
        t = rng.integers(0, 256, size=(411, nb), dtype=np.uint8)
        t[100:150] = q[:50]
        idx1, d1, d2 = np.empty(300, np.int32), np.empty(300, np.uint16), np.empty(300, np.uint16)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        ctx.check(hip.lib.gh_bf_match_bytes_host(ctx.h, p(q), 300, p(t), 411, nb, p(idx1), p(d1), p(d2)))
        e = oracle.bf_match_bytes(q, t, nb)
        assert np.array_equal(idx1, e[0]) and np.array_equal(d1, e[1]) and np.array_equal(d2, e[2])


@pytest.mark.parametrize("nb", [64, 24])
def test_wide_descriptors_train_set_beyond_65535_rows(ctx, oracle, nb):
    """Round 6: the wide path chunks large train sets like the 32-byte one (it refused nt > 65535 in round 5): duplicates of the
    best row in later chunks do not displace the first one, a strictly better row in the last chunk wins, duplicates across
    the chunk boundary keep the earlier index.  Device and host entries against the oracle (pinned to the reference's hamming64 /
    hamming8x)."""
    import ctypes as C
    import torch
    from gslam_amd import hip
    from gslam_amd.matcher import BFMatcher
    rng = np.random.default_rng(nb)
    nq, nt = 200, 140000
    q = rng.integers(0, 256, (nq, nb), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, nb), dtype=np.uint8)
    t[1000] = q[0]; t[70000] = q[0]; t[135000] = q[0]
    t[66000] = q[1]; t[66000, 0] ^= 1; t[139999] = q[1]
    t[65531] = q[2]; t[65532] = q[2]
    e = oracle.bf_match_bytes(q, t, nb)
    assert e[0][0] == 1000 and e[2][0] == 0 and e[0][1] == 139999 and e[2][1] == 1 and e[0][2] == 65531
    m = BFMatcher(ctx)
    idx1, d1, d2 = m.match_bytes(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(idx1.cpu().numpy(), e[0])
    assert np.array_equal(d1.cpu().numpy().view(np.uint16), e[1]) and np.array_equal(d2.cpu().numpy().view(np.uint16), e[2])
    hi, h1, h2 = np.empty(nq, np.int32), np.empty(nq, np.uint16), np.empty(nq, np.uint16)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    ctx.check(hip.lib.gh_bf_match_bytes_host(ctx.h, p(q), nq, p(t), nt, nb, p(hi), p(h1), p(h2)))
    assert np.array_equal(hi, e[0]) and np.array_equal(h1, e[1]) and np.array_equal(h2, e[2])
