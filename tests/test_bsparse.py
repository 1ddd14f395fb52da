"""Block-sparse pose-graph solver (gslam_amd/csrc/bsparse.hip).  CPU part: the symbolic factorisation (gh_bs_symbolic is
host code) -- the order is a permutation, every round is an independent set, and the block pattern it predicts is
COMPLETE: a numpy block elimination restricted to that pattern reproduces the dense solve (a missing fill block would
not).  GPU part: the numeric factorisation against numpy, and the LM loop through it against the oracle / the dense path."""
import numpy as np
import pytest

from gslam_amd import posegraph


def chain_with_loops(nf, n_loops, seed):
    rng = np.random.default_rng(seed)
    pairs = {(i + 1, i) for i in range(nf - 1)}
    while len(pairs) < nf - 1 + n_loops:
        a, b = sorted(rng.integers(0, nf, 2).tolist())
        if b - a > 1:
            pairs.add((b, a))
    pairs = sorted(pairs)
    return np.array([p[0] for p in pairs], np.int32), np.array([p[1] for p in pairs], np.int32)


def spd_blocks(nf, prow, pcol, seed):
    """Random symmetric positive definite block matrix on the pattern (diagonally dominant through J^T J terms)."""
    rng = np.random.default_rng(seed)
    diag = np.zeros((nf, 7, 7))
    off = np.zeros((len(prow), 7, 7))
    for k, (r, c) in enumerate(zip(prow, pcol)):
        Jr, Jc = rng.standard_normal((7, 7)), rng.standard_normal((7, 7))
        diag[r] += Jr.T @ Jr
        diag[c] += Jc.T @ Jc
        off[k] = Jr.T @ Jc  # block (r, c)
    for f in range(nf):
        diag[f] += 1e-3 * np.eye(7)
    return diag, off


def dense_of(nf, prow, pcol, diag, off):
    H = np.zeros((7 * nf, 7 * nf))
    for f in range(nf):
        H[7 * f:7 * f + 7, 7 * f:7 * f + 7] = diag[f]
    for k, (r, c) in enumerate(zip(prow, pcol)):
        H[7 * r:7 * r + 7, 7 * c:7 * c + 7] += off[k]
        H[7 * c:7 * c + 7, 7 * r:7 * r + 7] += off[k].T
    return H


def to_colmajor(blocks):
    return np.ascontiguousarray(np.transpose(blocks, (0, 2, 1))).reshape(len(blocks), 49)


@pytest.mark.parametrize("nf,loops,root_min", [(40, 0, 4), (200, 12, 16), (600, 40, 64), (64, 200, 8)])
def test_symbolic_order_rounds_and_pattern(nf, loops, root_min):
    prow, pcol = chain_with_loops(nf, loops, 7)
    sym = posegraph.bs_symbolic(nf, prow, pcol, root_min=root_min)
    pos, ns, nr = sym["pos"], sym["ns"], sym["nr"]
    assert sorted(pos.tolist()) == list(range(nf)) and ns + nr == nf and nr >= min(root_min, nf)
    colptr, rows, rp = sym["colptr"], sym["rows"], sym["round_ptr"]
    assert rp[0] == 0 and rp[-1] == ns and np.all(np.diff(rp) > 0)
    rnd = np.zeros(nf, np.int64) + len(rp)
    for r in range(len(rp) - 1):
        rnd[rp[r]:rp[r + 1]] = r
    for c in range(ns):
        rc = rows[colptr[c]:colptr[c + 1]]
        assert np.all(np.diff(rc) > 0) and (len(rc) == 0 or rc[0] > c)
        assert np.all(rnd[rc] > rnd[c])  # a round is an independent set: a column only reaches later rounds / the root
    if loops == 0:
        assert len(rp) - 1 <= int(np.ceil(np.log2(nf / root_min))) + 2  # cyclic reduction on a chain

    # numeric elimination on the predicted pattern only == dense solve
    diag, off = spd_blocks(nf, prow, pcol, 3)
    H = dense_of(nf, prow, pcol, diag, off)
    perm = np.argsort(pos)  # position -> frame
    idx = (7 * perm[:, None] + np.arange(7)[None, :]).reshape(-1)
    A = H[np.ix_(idx, idx)].copy()
    pattern = np.zeros((nf, nf), bool)
    for c in range(ns):
        pattern[rows[colptr[c]:colptr[c + 1]], c] = True
    pattern[ns:, ns:] = True
    # every original block is inside the pattern
    for r, c in zip(prow, pcol):
        a, b = max(pos[r], pos[c]), min(pos[r], pos[c])
        assert pattern[a, b]
    mask = np.kron(np.tril(pattern, -1) | np.eye(nf, dtype=bool), np.ones((7, 7), bool))
    L = np.zeros_like(A)
    for c in range(nf):  # right-looking block Cholesky that DROPS anything outside the pattern
        s = slice(7 * c, 7 * c + 7)
        L[s, s] = np.linalg.cholesky(A[s, s])
        below = slice(7 * c + 7, 7 * nf)
        L[below, s] = np.linalg.solve(L[s, s], A[below, s].T).T * mask[below, s]
        A[below, below] -= (L[below, s] @ L[below, s].T) * mask[below, below]
    b = np.random.default_rng(5).standard_normal(7 * nf)
    x = np.linalg.solve(L.T, np.linalg.solve(L, b[idx]))
    ref = np.linalg.solve(H, b)[idx]
    assert np.max(np.abs(x - ref)) <= 1e-8 * np.max(np.abs(ref))


def test_symbolic_degenerate_graphs():
    # no edges at all: everything beyond the root minimum goes in one round
    sym = posegraph.bs_symbolic(50, np.zeros(0, np.int32), np.zeros(0, np.int32), root_min=10)
    assert sym["ns"] == 40 and len(sym["round_ptr"]) == 2 and len(sym["rows"]) == 0
    # a clique: nothing is worth eliminating
    n = 30
    pr, pc = np.array([(a, b) for a in range(n) for b in range(a)], np.int32).T
    sym = posegraph.bs_symbolic(n, pr, pc, root_min=4)
    assert sym["ns"] == 0 and sym["nr"] == n
    # a star: the leaves go first, the hub stays
    pr, pc = np.arange(1, 100, dtype=np.int32), np.zeros(99, np.int32)
    sym = posegraph.bs_symbolic(100, pr, pc, root_min=1)
    assert sym["pos"][0] == 99


@pytest.mark.gpu
@pytest.mark.parametrize("nf,loops,root_min", [(12, 0, 2), (300, 20, 16), (2000, 150, 128), (64, 300, 8), (50, 0, 50)])
def test_numeric_factorisation_against_numpy(nf, loops, root_min):
    from gslam_amd import hip
    ctx = hip.Context()
    prow, pcol = chain_with_loops(nf, loops, 11)
    diag, off = spd_blocks(nf, prow, pcol, 13)
    H = dense_of(nf, prow, pcol, diag, off)
    g = np.random.default_rng(17).standard_normal(7 * nf)
    radius = 50.0
    D = np.clip(np.diag(H), 1e-6, 1e32) / radius
    ref = np.linalg.solve(H + np.diag(D), -g)
    x, info = posegraph.bs_solve(ctx, prow, pcol, to_colmajor(diag), to_colmajor(off), g, radius=radius, root_min=root_min)
    assert info == 0
    assert np.max(np.abs(x - ref)) <= 1e-9 * np.max(np.abs(ref))
    x2, _ = posegraph.bs_solve(ctx, prow, pcol, to_colmajor(diag), to_colmajor(off), g, radius=radius, root_min=root_min)
    assert x2.tobytes() == x.tobytes()  # no atomics: bitwise reproducible


@pytest.mark.gpu
def test_numeric_factorisation_reports_a_bad_pivot():
    from gslam_amd import hip
    ctx = hip.Context()
    nf = 40
    prow, pcol = chain_with_loops(nf, 0, 1)
    diag, off = spd_blocks(nf, prow, pcol, 2)
    diag[7] = -np.eye(7)
    x, info = posegraph.bs_solve(ctx, prow, pcol, to_colmajor(diag), to_colmajor(off), np.ones(7 * nf), radius=1e30, root_min=4)
    assert info != 0


def _random_graph(rng, nf, density):
    pairs = set()
    for a in range(nf):
        for b in range(a):
            if rng.random() < density:
                pairs.add((a, b))
    if not pairs:
        pairs.add((nf - 1, 0))
    pairs = sorted(pairs)
    return np.array([p[0] for p in pairs], np.int32), np.array([p[1] for p in pairs], np.int32)


def test_symbolic_fuzz_random_graphs():
    """Arbitrary keyframe graphs (trees, dense clusters, disconnected parts): positions are a permutation, rounds are
    independent sets, and the predicted pattern contains the exact fill of the elimination order."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=60, deadline=None)
    @given(st.integers(2, 36), st.floats(0.02, 0.6), st.integers(1, 12), st.integers(0, 2**31 - 1))
    def run(nf, density, root_min, seed):
        rng = np.random.default_rng(seed)
        prow, pcol = _random_graph(rng, nf, density)
        sym = posegraph.bs_symbolic(nf, prow, pcol, root_min=root_min)
        pos, ns = sym["pos"], sym["ns"]
        assert sorted(pos.tolist()) == list(range(nf))
        assert sym["nr"] >= min(root_min, nf)
        colptr, rows, rp = sym["colptr"], sym["rows"], sym["round_ptr"]
        # exact symbolic elimination of the permuted graph (boolean): every fill block below the diagonal of a sparse
        # column must be in the pattern, and nothing else may be
        A = np.zeros((nf, nf), bool)
        A[pos[prow], pos[pcol]] = True
        A |= A.T
        for c in range(ns):
            nb = np.nonzero(A[c + 1:, c])[0] + c + 1
            assert rows[colptr[c]:colptr[c + 1]].tolist() == nb.tolist()
            A[np.ix_(nb, nb)] = True
            np.fill_diagonal(A, False)
        rnd = np.full(nf, len(rp))
        for r in range(len(rp) - 1):
            rnd[rp[r]:rp[r + 1]] = r
        for c in range(ns):
            assert np.all(rnd[rows[colptr[c]:colptr[c + 1]]] > rnd[c])

    run()


@pytest.mark.gpu
def test_numeric_fuzz_random_graphs():
    from hypothesis import given, settings, strategies as st
    from gslam_amd import hip
    ctx = hip.Context()

    @settings(max_examples=40, deadline=None)
    @given(st.integers(2, 60), st.floats(0.02, 0.5), st.integers(1, 16), st.integers(0, 2**31 - 1))
    def run(nf, density, root_min, seed):
        rng = np.random.default_rng(seed)
        prow, pcol = _random_graph(rng, nf, density)
        diag, off = spd_blocks(nf, prow, pcol, seed % 1000)
        H = dense_of(nf, prow, pcol, diag, off)
        g = rng.standard_normal(7 * nf)
        radius = 10.0 ** rng.uniform(0, 6)
        ref = np.linalg.solve(H + np.diag(np.clip(np.diag(H), 1e-6, 1e32) / radius), -g)
        x, info = posegraph.bs_solve(ctx, prow, pcol, to_colmajor(diag), to_colmajor(off), g, radius=radius, root_min=root_min)
        assert info == 0
        assert np.max(np.abs(x - ref)) <= 1e-8 * max(np.max(np.abs(ref)), 1e-30)

    run()
