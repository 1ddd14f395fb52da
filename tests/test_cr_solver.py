"""Band SPD solve by block cyclic reduction (gslam_amd/csrc/chol_cr.hip).

CPU part: an independent numpy restatement of the elimination schedule the kernels follow (levels of stride 2^r, the
two neighbours of an eliminated superblock, the fill block, the backward pass) checked against numpy's dense solve --
it pins the ALGORITHM without a GPU.  GPU part: gh_band_solve_dev against numpy on band matrices of every tile count,
ragged last superblocks, and against the dense gh_potrf_solve_dev path.
Reference anchor: the linear solve inside Optimizer::optimize (GSLAM/core/Optimizer.h:229; Ceres SPARSE_SCHUR restated).
"""
import numpy as np
import pytest


def make_band(n, hb, seed=0, cond_boost=1.0):
    rng = np.random.default_rng(seed)
    A = np.zeros((n, n))
    for i in range(n):
        lo = max(0, i - hb)
        A[i, lo:i + 1] = rng.standard_normal(i - lo + 1)
    A = np.tril(A)
    A = A + A.T
    A += np.eye(n) * (np.abs(A).sum(1).max() * cond_boost + 1.0)
    return A


def cr_solve_restated(S, b, m):
    """Block cyclic reduction on superblocks of m columns (half-bandwidth of S <= m), as chol_cr.hip schedules it."""
    n = S.shape[0]
    N = -(-n // m)
    A = np.tril(S).copy()
    rhs = b.copy()
    blk = lambda k: slice(k * m, min((k + 1) * m, n))
    W, L, levels = {}, {}, []
    s = 1
    while s < N:
        elim = list(range(s, N, 2 * s))
        levels.append((s, elim))
        for i in elim:
            D = A[blk(i), blk(i)]
            L[i] = np.linalg.cholesky(np.tril(D) + np.tril(D, -1).T)
        for i in elim:
            u, d = i - s, i + s
            if u >= 0:
                W[(i, 0)] = np.linalg.solve(L[i], A[blk(i), blk(u)]).T      # B(i,u)^T L^-T
            if d < N:
                W[(i, 1)] = np.linalg.solve(L[i], A[blk(d), blk(i)].T).T    # B(d,i) L^-T
            rhs[blk(i)] = np.linalg.solve(L[i], rhs[blk(i)])
        for i in elim:
            u, d = i - s, i + s
            y = rhs[blk(i)]
            if u >= 0:
                A[blk(u), blk(u)] -= np.tril(W[(i, 0)] @ W[(i, 0)].T)
                rhs[blk(u)] -= W[(i, 0)] @ y
            if d < N:
                A[blk(d), blk(d)] -= np.tril(W[(i, 1)] @ W[(i, 1)].T)
                rhs[blk(d)] -= W[(i, 1)] @ y
            if u >= 0 and d < N:
                A[blk(d), blk(u)] -= W[(i, 1)] @ W[(i, 0)].T
        s *= 2
    D = A[blk(0), blk(0)]
    L0 = np.linalg.cholesky(np.tril(D) + np.tril(D, -1).T)
    x = np.zeros(n)
    x[blk(0)] = np.linalg.solve(L0.T, np.linalg.solve(L0, rhs[blk(0)]))
    for s, elim in reversed(levels):
        for i in elim:
            t = rhs[blk(i)].copy()
            if i - s >= 0:
                t -= W[(i, 0)].T @ x[blk(i - s)]
            if i + s < N:
                t -= W[(i, 1)].T @ x[blk(i + s)]
            x[blk(i)] = np.linalg.solve(L[i].T, t)
    return x


@pytest.mark.parametrize("n,hb,m", [(3000, 149, 192), (1000, 60, 64), (777, 100, 128), (500, 191, 192), (200, 63, 64),
                                    (1345, 128, 128)])
def test_schedule_restated_matches_dense(n, hb, m):
    S = make_band(n, hb, seed=n)
    b = np.random.default_rng(1).standard_normal(n)
    x = cr_solve_restated(S, b, m)
    xr = np.linalg.solve(S, b)
    assert np.abs(x - xr).max() <= 1e-13 * np.abs(xr).max()


def make_arrow(n_band, hb, nbr, seed=0, fill=1.0):
    """SPD arrowhead: a band of n_band unknowns + nbr dense border rows (fill = fraction of non-zero border columns)."""
    rng = np.random.default_rng(seed)
    n = n_band + nbr
    A = np.zeros((n, n))
    for i in range(n_band):
        lo = max(0, i - hb)
        A[i, lo:i + 1] = rng.standard_normal(i - lo + 1)
    E = rng.standard_normal((nbr, n_band))
    if fill < 1.0:
        E *= rng.random((nbr, n_band)) < fill
    A[n_band:, :n_band] = E
    A[n_band:, n_band:] = np.tril(rng.standard_normal((nbr, nbr)))
    A = np.tril(A)
    A = A + A.T
    A += np.eye(n) * (np.abs(A).sum(1).max() + 1.0)
    return A


def arrow_solve_restated(S, b, n_band, m, top=1, on_eliminate=None):
    """The schedule of chol_cr.hip's arrowhead mode: block cyclic reduction on the band with the border rows E and the
    right-hand side riding along as extra rows (Y_i = E_i L_i^-T, E_u -= Y_i W_u^T, E_d -= Y_i W_d^T), the corner update
    C -= sum_i Y_i Y_i^T over every eliminated superblock, the dense solve of superblock 0 + border, and the backward pass
    with - x_c Y_i added to the right-hand side of superblock i.
    top > 1 (the DENSE TOP of cr_solve_t): the reduction stops as soon as at most `top` superblocks survive (0, st, 2 st, ...);
    they -- block tridiagonal among themselves -- join the border in the dense system instead of superblock 0 alone."""
    n = S.shape[0]
    nbr = n - n_band
    N = -(-n_band // m)
    A = np.tril(S).copy()
    rhs = b.copy()
    blk = lambda k: slice(k * m, min((k + 1) * m, n_band))
    bord = slice(n_band, n)
    W, L, Y, levels = {}, {}, {}, []
    st = 1
    while -(-N // st) > top:
        st *= 2
    surv = list(range(0, N, st))
    s = 1
    while s < st:
        elim = list(range(s, N, 2 * s))
        levels.append((s, elim))
        for i in elim:
            D = A[blk(i), blk(i)]
            L[i] = np.linalg.cholesky(np.tril(D) + np.tril(D, -1).T)
            u, d = i - s, i + s
            if u >= 0:
                W[(i, 0)] = np.linalg.solve(L[i], A[blk(i), blk(u)]).T
            if d < N:
                W[(i, 1)] = np.linalg.solve(L[i], A[blk(d), blk(i)].T).T
            rhs[blk(i)] = np.linalg.solve(L[i], rhs[blk(i)])
            if on_eliminate is not None:
                on_eliminate(i, A[bord, blk(i)])
            Y[i] = np.linalg.solve(L[i], A[bord, blk(i)].T).T            # E_i L_i^-T  (nbr x m)
            A[bord, blk(i)] = Y[i]
        for i in elim:
            u, d = i - s, i + s
            y = rhs[blk(i)]
            if u >= 0:
                A[blk(u), blk(u)] -= np.tril(W[(i, 0)] @ W[(i, 0)].T)
                rhs[blk(u)] -= W[(i, 0)] @ y
                A[bord, blk(u)] -= Y[i] @ W[(i, 0)].T
            if d < N:
                A[blk(d), blk(d)] -= np.tril(W[(i, 1)] @ W[(i, 1)].T)
                rhs[blk(d)] -= W[(i, 1)] @ y
                A[bord, blk(d)] -= Y[i] @ W[(i, 1)].T
            if u >= 0 and d < N:
                A[blk(d), blk(u)] -= W[(i, 1)] @ W[(i, 0)].T
        s *= 2
    # corner: every eliminated superblock at once (their columns of the border rows now hold the Y_i)
    ecols = np.concatenate([np.arange(n_band)[blk(i)] for i in range(N) if i not in surv]) if len(surv) < N else np.zeros(0, int)
    Yall = A[bord][:, ecols]
    C = A[bord, bord] - np.tril(Yall @ Yall.T)
    gc = rhs[bord] - Yall @ rhs[ecols]
    # dense system of the survivors (D_j, B(j + st, j), nothing else) + border
    scols = np.concatenate([np.arange(n_band)[blk(j)] for j in surv])
    qb = len(scols)
    q = qb + nbr
    M = np.zeros((q, q))
    off = np.cumsum([0] + [len(np.arange(n_band)[blk(j)]) for j in surv])
    for a_, j in enumerate(surv):
        M[off[a_]:off[a_ + 1], off[a_]:off[a_ + 1]] = A[blk(j), blk(j)]
        if a_ + 1 < len(surv):
            M[off[a_ + 1]:off[a_ + 2], off[a_]:off[a_ + 1]] = A[blk(surv[a_ + 1]), blk(j)]
        M[qb:, off[a_]:off[a_ + 1]] = A[bord, blk(j)]
    M[qb:, qb:] = C
    M = np.tril(M) + np.tril(M, -1).T
    xq = np.linalg.solve(M, np.concatenate([rhs[scols], gc]))
    x = np.zeros(n)
    x[scols] = xq[:qb]
    x[bord] = xq[qb:]
    for s, elim in reversed(levels):
        for i in elim:
            t = rhs[blk(i)] - Y[i].T @ x[bord]
            if i - s >= 0:
                t -= W[(i, 0)].T @ x[blk(i - s)]
            if i + s < N:
                t -= W[(i, 1)].T @ x[blk(i + s)]
            x[blk(i)] = np.linalg.solve(L[i].T, t)
    return x


@pytest.mark.parametrize("n_band,hb,m,nbr", [(3000, 149, 192, 90), (1000, 60, 64, 7), (777, 100, 128, 130), (520, 191, 192, 64),
                                             (1345, 128, 128, 1)])
def test_arrow_schedule_restated_matches_dense(n_band, hb, m, nbr):
    S = make_arrow(n_band, hb, nbr, seed=n_band + nbr)
    b = np.random.default_rng(1).standard_normal(n_band + nbr)
    x = arrow_solve_restated(S, b, n_band, m)
    xr = np.linalg.solve(S, b)
    assert np.abs(x - xr).max() <= 1e-13 * np.abs(xr).max()


@pytest.mark.parametrize("n_band,hb,m,nbr,top", [(3000, 149, 192, 90, 4), (3000, 149, 192, 0, 4), (1000, 60, 64, 7, 4), (777, 100, 128, 130, 2),
                                                 (520, 191, 192, 64, 4), (1345, 128, 128, 0, 8), (1345, 128, 128, 1, 3), (1000, 60, 64, 0, 16)])
def test_dense_top_schedule_restated_matches_dense(n_band, hb, m, nbr, top):
    """The reduction stopped at `top` survivors (chol_cr.hip: DENSE TOP, GSLAM_HIP_CR_TOP), with and without a border; a last
    survivor that is a partial superblock and a `top` that is not a power of two included."""
    S = make_arrow(n_band, hb, nbr, seed=n_band + nbr) if nbr else make_band(n_band, hb, seed=n_band)
    b = np.random.default_rng(1).standard_normal(n_band + nbr)
    x = arrow_solve_restated(S, b, n_band, m, top=top)
    xr = np.linalg.solve(S, b)
    assert np.abs(x - xr).max() <= 1e-13 * np.abs(xr).max()


def _load_lib_cpu():
    """the C-ABI library without a GPU context (host-only entry points)"""
    from gslam_amd import hip
    return hip.lib


@pytest.mark.parametrize("n_band,hb,m,nbr,top", [(3000, 149, 192, 90, 4), (2000, 60, 64, 130, 4), (1345, 128, 128, 40, 4), (3000, 149, 192, 300, 4)])
def test_border_structure_is_conservative(n_band, hb, m, nbr, top):
    """gh_cr_border_structure (what lets the border kernels of the arrowhead solver skip blocks): on a border that couples every
    16-row strip to a few random places of the band, every (superblock, strip) block that is non-zero in the numpy restatement
    at the moment the superblock is eliminated is marked -- and the marking is not trivial (most blocks stay unmarked)."""
    import ctypes as C
    lib = _load_lib_cpu()
    rng = np.random.default_rng(n_band + nbr)
    S = make_arrow(n_band, hb, nbr, seed=n_band + nbr)
    N, nbs = -(-n_band // m), -(-nbr // 16)
    E = np.zeros((nbr, n_band))
    init = np.zeros((N, nbs), np.uint8)
    for t in range(nbs):
        for c0 in rng.integers(0, n_band - 12, 3):  # three "cameras" of 6 columns per strip, two strips may share one
            r0, r1 = 16 * t, min(nbr, 16 * t + 16)
            E[r0:r1, c0:c0 + 6] = rng.standard_normal((r1 - r0, 6)) * 0.05
            init[c0 // m, t] = 1
            init[(c0 + 5) // m, t] = 1
    S[n_band:, :n_band] = E
    S[:n_band, n_band:] = E.T
    S[n_band:, n_band:] += np.eye(nbr) * 2.0
    need = lib.gh_cr_border_structure(n_band, m // 64, nbr, None, None, 0)
    ntr = -(-(nbr + 1) // 64)
    assert need == N * (nbs + ntr)
    out = np.zeros(need, np.uint8)
    assert lib.gh_cr_border_structure(n_band, m // 64, nbr, init.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), need) == need
    nzY, nzT = out[:N * nbs].reshape(N, nbs), out[N * nbs:].reshape(N, ntr)
    seen = {}

    def on_eliminate(i, Ei):
        seen[i] = np.array([np.any(Ei[16 * t:16 * t + 16] != 0.0) for t in range(nbs)])

    b = rng.standard_normal(n_band + nbr)
    x = arrow_solve_restated(S, b, n_band, m, top=top, on_eliminate=on_eliminate)
    assert np.abs(x - np.linalg.solve(S, b)).max() <= 1e-12 * np.abs(x).max()
    assert len(seen) >= N - top
    for i, nz in seen.items():
        assert not np.any(nz & (nzY[i] == 0)), "superblock %d: a non-zero strip is marked zero" % i
        for t in range(ntr):
            assert nzT[i, t] == (1 if (t == nbr // 64 or nzY[i, 4 * t:4 * t + 4].any()) else 0)
    elim = sorted(seen)
    assert nzY[elim].mean() < 0.6  # (sparse input: most blocks are skipped)
    surv = [i for i in range(N) if i not in seen]
    assert nzY[surv].all()


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def ctx():
    import torch
    from gslam_amd import hip
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return hip.Context(0, stream=torch.cuda.current_stream().cuda_stream)


@pytest.mark.gpu
@pytest.mark.parametrize("n,hb", [(3000, 149), (3000, 191), (1024, 64), (1000, 40), (777, 100), (1345, 128), (260, 17),
                                  (3005, 150), (768, 192), (6000, 149)])
def test_band_solve_vs_numpy(ctx, n, hb):
    from gslam_amd import ba
    S = make_band(n, hb, seed=n + hb)
    b = np.random.default_rng(2).standard_normal(n)
    x, info = ba.band_solve(ctx, S, b, hb)
    assert info == 0
    xr = np.linalg.solve(S, b)
    assert np.abs(x - xr).max() <= 1e-12 * np.abs(xr).max()
    # and it is the factorisation the restatement describes: same numbers to rounding
    m = 64 * (-(-hb // 64))
    xs = cr_solve_restated(S, b, m)
    assert np.abs(x - xs).max() <= 1e-12 * np.abs(xr).max()


@pytest.mark.gpu
@pytest.mark.parametrize("n_band,hb,nbr,fill", [(3000, 149, 90, 1.0), (3000, 191, 360, 0.05), (1024, 64, 7, 1.0), (1000, 40, 1, 1.0),
                                                (777, 100, 130, 0.3), (1345, 128, 65, 1.0), (260, 17, 300, 1.0),
                                                (3005, 150, 17, 1.0), (768, 192, 64, 1.0), (6000, 149, 128, 0.02)])
def test_arrow_solve_vs_numpy(ctx, n_band, hb, nbr, fill):
    """gh_arrow_solve_dev (band + dense border) against numpy and against the restated schedule: every tile count, ragged
    last superblocks, borders from one row to wider than the band part's superblock, sparse and dense border rows."""
    from gslam_amd import ba
    S = make_arrow(n_band, hb, nbr, seed=n_band + hb + nbr, fill=fill)
    b = np.random.default_rng(2).standard_normal(n_band + nbr)
    x, info = ba.arrow_solve(ctx, S, b, n_band, hb)
    assert info == 0
    xr = np.linalg.solve(S, b)
    assert np.abs(x - xr).max() <= 1e-12 * np.abs(xr).max()
    m = 64 * (-(-hb // 64))
    xs = arrow_solve_restated(S, b, n_band, m)
    assert np.abs(x - xs).max() <= 1e-12 * np.abs(xr).max()


@pytest.mark.gpu
@pytest.mark.parametrize("n_band,hb,nbr,fill", [(3000, 149, 90, 1.0), (3000, 149, 0, 1.0), (876, 149, 84, 0.3), (954, 149, 6, 1.0), (768, 192, 64, 1.0),
                                                (1024, 64, 7, 1.0), (1000, 40, 0, 1.0), (777, 100, 130, 0.3), (1345, 128, 65, 1.0), (260, 17, 300, 1.0),
                                                (6000, 149, 128, 0.02), (1620, 149, 180, 0.5), (590, 191, 33, 1.0)])
def test_arrow_solve_compact_layout_vs_numpy(ctx, n_band, hb, nbr, fill):
    """The compact columns gh_ba_solve keeps its reduced camera system in (cr_map.h; gh_cr_compact_layout): the same systems as
    test_arrow_solve_vs_numpy, laid out compactly on the host, against numpy and against the dense-layout entry -- every tile count,
    1 .. 5 reduction levels (one fill slot each), partial last superblocks, with and without a border."""
    from gslam_amd import ba
    S = make_arrow(n_band, hb, nbr, seed=n_band + hb + nbr, fill=fill) if nbr else make_band(n_band, hb, seed=n_band + hb)
    b = np.random.default_rng(2).standard_normal(n_band + nbr)
    x, info = ba.arrow_solve_compact(ctx, S, b, n_band, hb)
    assert info == 0
    xr = np.linalg.solve(S, b)
    assert np.abs(x - xr).max() <= 1e-12 * np.abs(xr).max()
    xd, info_d = (ba.arrow_solve(ctx, S, b, n_band, hb) if nbr else ba.band_solve(ctx, S, b, hb))
    assert info_d == 0 and np.array_equal(x, xd), "the two layouts run the same arithmetic"


@pytest.mark.gpu
def test_arrow_solve_border_around_the_single_launch_threshold(ctx):
    """the workspace was sized with the LARGEST dense-top system (top * m + nbr unknowns) while the solve asks
    for the single-launch factorisation's state with the ACTUAL one (fewer / partial survivors); the state size drops to 0 once a
    shape exceeds the CU count, so in between the flow state lay past the reserved block.  n_band = 900 at m = 192: five
    superblocks, three survive (0, 2, 4; the last one 132 wide) -> 516 band unknowns in the dense top against 768 reserved for.
    Border widths on both sides of, and inside, that window for this device's CU count."""
    from gslam_amd import ba
    cus = ctx.device_info()["cu_count"]

    def fits(qn):  # chol.hip flow_groups, FL_MAXT = 6
        ntr = -(-(qn + 1) // 64)
        return 1 + max(ntr - 1, 0) + sum(-(-(i - 1) // 6) for i in range(2, ntr)) <= cus
    qn_last = max(q for q in range(64, 8192) if fits(q))
    n_band, hb, qb_actual, qb_reserved = 900, 149, 516, 768
    widths = sorted({qn_last - qb_reserved - 40, qn_last - qb_reserved + 1, qn_last - (qb_actual + qb_reserved) // 2,
                     qn_last - qb_actual, qn_last - qb_actual + 1})
    for nbr in widths:
        S = make_arrow(n_band, hb, nbr, seed=nbr, fill=0.2)
        b = np.random.default_rng(3).standard_normal(n_band + nbr)
        for rep in range(2):
            x, info = ba.arrow_solve(ctx, S, b, n_band, hb)
            assert info == 0, nbr
            xr = np.linalg.solve(S, b)
            assert np.abs(x - xr).max() <= 1e-12 * np.abs(xr).max(), nbr


@pytest.mark.gpu
def test_arrow_solve_is_reproducible_and_equals_dense_path(ctx):
    from gslam_amd import ba
    S = make_arrow(3000, 149, 200, seed=11, fill=0.1)
    b = np.random.default_rng(3).standard_normal(3200)
    x1, info1 = ba.arrow_solve(ctx, S, b, 3000, 149)
    x2, info2 = ba.arrow_solve(ctx, S, b, 3000, 149)
    _, xd, infod = ba.potrf_solve(ctx, S, b)
    assert info1 == 0 and info2 == 0 and infod == 0
    assert x1.tobytes() == x2.tobytes(), "fixed summation order: two runs agree bit for bit"
    assert np.abs(x1 - xd).max() <= 1e-12 * np.abs(xd).max()


@pytest.mark.gpu
def test_band_solve_ill_conditioned_residual(ctx):
    """A weakly dominant band (condition ~1e6): the residual, not the error, is the bar."""
    from gslam_amd import ba
    n, hb = 3000, 149
    rng = np.random.default_rng(5)
    G = np.zeros((n, n))
    for i in range(n):
        lo = max(0, i - hb // 2)
        G[i, lo:i + 1] = rng.standard_normal(i - lo + 1)
    S = G @ G.T + 1e-3 * np.eye(n)   # half-bandwidth hb - 1 at most, SPD
    assert np.abs(np.tril(S, -hb - 1)).max() == 0.0
    b = rng.standard_normal(n)
    x, info = ba.band_solve(ctx, S, b, hb)
    assert info == 0
    r = S @ x - b
    xr = np.linalg.solve(S, b)
    rr = S @ xr - b
    assert np.linalg.norm(r) <= 10 * max(np.linalg.norm(rr), 1e-13 * np.linalg.norm(b))


@pytest.mark.gpu
def test_band_solve_equals_dense_path(ctx):
    from gslam_amd import ba
    n, hb = 3000, 149
    S = make_band(n, hb, seed=9)
    b = np.random.default_rng(3).standard_normal(n)
    x, info = ba.band_solve(ctx, S, b, hb)
    _, xd, infod = ba.potrf_solve(ctx, S, b)
    assert info == 0 and infod == 0
    assert np.abs(x - xd).max() <= 1e-12 * np.abs(xd).max()


@pytest.mark.gpu
@pytest.mark.parametrize("where", [500, 5, 520, 999])
def test_band_solve_not_positive_definite(ctx, where):
    """A negative pivot in a superblock the reduction eliminates (500, 999) and in one that survives into the dense top (5, 520:
    superblocks 0 and 8 of 16): reported as a column of THIS system, never as the code of an expired wait (info > n)."""
    from gslam_amd import ba
    n, hb = 1000, 60
    S = make_band(n, hb, seed=4)
    S[where, where] = -1.0
    _, info = ba.band_solve(ctx, S, np.ones(n), hb)
    assert 1 <= info <= n


@pytest.mark.gpu
def test_band_solve_refuses_wide_band(ctx):
    from gslam_amd import ba, hip
    S = make_band(1000, 10, seed=1)
    with pytest.raises(hip.GslamHipError):
        ba.band_solve(ctx, S, np.ones(1000), 200)      # beyond three tiles
    with pytest.raises(hip.GslamHipError):
        ba.band_solve(ctx, S[:100, :100], np.ones(100), 60)   # fewer than four superblocks


# ------------------------------------------------------------------------------------------------- inside the LM loop
def _solve_with(ctx, g, solver, iters=30):
    from gslam_amd import ba
    ctx.set_ba_solver(solver)
    try:
        poses, pts, s, st = ba.solve(ctx, g, ba.default_options(max_iterations=iters))
        used = ctx.last_ba_solver()
    finally:
        ctx.set_ba_solver("auto")
    return poses, pts, s, used


@pytest.mark.gpu
@pytest.mark.parametrize("cams,points", [(160, 8000), (500, 50000)])
def test_ba_band_and_dense_solvers_agree(ctx, cams, points):
    """Same graph through both linear solvers: identical LM decisions, costs to 1e-10, states to 1e-9 (C4 at full size
    included).  The graphs of ba_synth are trajectories: every point is seen from cameras at most 24 indices apart."""
    from gslam_amd.ba_synth import make_graph
    g = make_graph(cams, points, n_obs_per_point=6, seed=2)
    pd, xd, sd, used_d = _solve_with(ctx, g, "dense")
    pb, xb, sb, used_b = _solve_with(ctx, g, "band")
    assert used_d[0] == "dense" and used_b[0] == "band" and used_b[1] == 3 and used_b[2] <= 32
    assert sd.iterations == sb.iterations and sd.trace_len == sb.trace_len
    n = sd.trace_len
    assert list(sd.trace_accepted[:n]) == list(sb.trace_accepted[:n])
    cd, cb = np.array(sd.trace_cost[:n]), np.array(sb.trace_cost[:n])
    assert np.abs(cd - cb).max() <= 1e-10 * np.abs(cd).max()
    assert np.abs(pd - pb).max() <= 1e-9 and np.abs(xd - xb).max() <= 1e-9
    # auto picks the band solver on this graph
    _, _, _, used_a = _solve_with(ctx, g, "auto", iters=1)
    assert used_a[0] == "band"


def _same_lm_run(sd, sb, pd, pb, xd, xb):
    assert sd.iterations == sb.iterations and sd.trace_len == sb.trace_len
    n = sd.trace_len
    assert list(sd.trace_accepted[:n]) == list(sb.trace_accepted[:n])
    cd, cb = np.array(sd.trace_cost[:n]), np.array(sb.trace_cost[:n])
    assert np.abs(cd - cb).max() <= 1e-10 * np.abs(cd).max()
    assert np.abs(pd - pb).max() <= 1e-9 and np.abs(xd - xb).max() <= 1e-9


@pytest.mark.gpu
def test_ba_one_loop_closure_takes_the_arrow_solver(ctx):
    """ONE observation from the other end of the trajectory used to send the whole graph to the dense factorisation (rounds
    1-4); now the camera it ties in joins the border of an arrowhead system.  Same LM run as the dense solver's."""
    from gslam_amd.ba_synth import make_graph
    g = make_graph(160, 8000, n_obs_per_point=6, seed=3)
    oc = np.array(g["obs_cam"]).copy()
    op = np.array(g["obs_point"])
    k = int(np.flatnonzero(op == op[0])[0])
    oc[k] = 159 if oc[k] < 80 else 0     # one observation from the other end of the trajectory
    g2 = dict(g)
    g2["obs_cam"] = oc
    pa, xa, sa, used = _solve_with(ctx, g2, "auto")
    assert used[0] == "arrow" and used[1] == 3 and used[2] <= 31 and sa.iterations >= 1
    pd, xd, sd, used_d = _solve_with(ctx, g2, "dense")
    assert used_d[0] == "dense"
    _same_lm_run(sd, sa, pd, pa, xd, xa)


@pytest.mark.gpu
@pytest.mark.parametrize("cams,points,closures,span", [(160, 8000, 5, 80), (300, 20000, 12, None), (500, 50000, 20, None)])
@pytest.mark.parametrize("structure", ["0", "1"])
def test_ba_arrow_and_dense_solvers_agree(ctx, monkeypatch, cams, points, closures, span, structure):
    """Trajectories with loop-closure points (ba_synth.make_graph(loop_closures=...)) through the dense factorisation and through
    the arrowhead solver (band + border, arrow ordering of the cameras inside gh_ba_solve): identical LM decisions, costs to
    1e-10, states to 1e-9 in the CALLER's camera order (C4 + 20 closures at full size included)."""
    from gslam_amd.ba_synth import make_graph
    # GSLAM_HIP_BA_ARROW_DENSE_BORDER = 0: the border kernels skip the blocks gh_cr_border_structure marks zero (by default only
    # from 4 M entries of border block on: C5-sized systems); 1: every block is processed
    monkeypatch.setenv("GSLAM_HIP_BA_ARROW_DENSE_BORDER", structure)
    g = make_graph(cams, points, n_obs_per_point=6, seed=2, loop_closures=closures, closure_span=span)
    pd, xd, sd, used_d = _solve_with(ctx, g, "dense")
    pa, xa, sa, used_a = _solve_with(ctx, g, "auto")
    assert used_d[0] == "dense" and used_a[0] == "arrow" and used_a[2] <= 31
    _same_lm_run(sd, sa, pd, pa, xd, xa)
    # the loop closures matter: without them the run is a different one
    g0 = make_graph(cams, points, n_obs_per_point=6, seed=2)
    _, _, s0, used_0 = _solve_with(ctx, g0, "auto")
    assert used_0[0] == "band" and s0.final_cost != sa.final_cost


@pytest.mark.gpu
@pytest.mark.parametrize("structure", ["0", "1"])
def test_ba_arrow_graph_session_round_trip(ctx, monkeypatch, structure):
    """The device-resident graph API keeps its cameras in arrow order inside: create / update / solve / read give the one-shot
    solve's result in the caller's camera order (with and without the border's block structure: it is kept with the session)."""
    from gslam_amd import ba
    from gslam_amd.ba_synth import make_graph
    monkeypatch.setenv("GSLAM_HIP_BA_ARROW_DENSE_BORDER", structure)
    g = make_graph(200, 10000, n_obs_per_point=6, seed=4, loop_closures=6)
    p1, x1, s1, _ = ba.solve(ctx, g, ba.default_options(max_iterations=15))
    assert ctx.last_ba_solver()[0] == "arrow"
    gr = ba.Graph(ctx, g, ba.default_options(max_iterations=15))
    try:
        s2, _ = gr.solve(ba.default_options(max_iterations=15))
        assert ctx.last_ba_solver()[0] == "arrow"
        p2, x2 = gr.read()
        assert s2.iterations == s1.iterations and abs(s2.final_cost - s1.final_cost) <= 1e-12 * s1.final_cost
        assert np.abs(p2 - p1).max() <= 1e-10 and np.abs(x2 - x1).max() <= 1e-10
        # a second solve from re-uploaded initial values (caller's order) reproduces it
        gr.update(cam_pose=g["cam_pose"], point_xyz=g["point_xyz"])
        s3, _ = gr.solve(ba.default_options(max_iterations=15))
        p3, x3 = gr.read()
        assert s3.iterations == s1.iterations and np.abs(p3 - p1).max() <= 1e-10 and np.abs(x3 - x1).max() <= 1e-10
    finally:
        gr.close()


@pytest.mark.gpu
def test_ba_wide_graph_stays_dense(ctx):
    """Observers drawn from the WHOLE trajectory for every point: the band that is left would be too short -> dense path, also
    when the band solver is asked for."""
    from gslam_amd.ba_synth import make_graph
    g = make_graph(160, 8000, n_obs_per_point=6, seed=3)
    rng = np.random.default_rng(0)
    oc = np.array(g["obs_cam"]).reshape(-1, 6).copy()
    for p in range(0, oc.shape[0], 2):
        oc[p] = np.sort(rng.choice(160, 6, replace=False))
    g2 = dict(g)
    g2["obs_cam"] = oc.reshape(-1).astype(np.int32)
    _, _, s, used = _solve_with(ctx, g2, "band", iters=3)
    assert used[0] == "dense" and used[2] > 31 and s.iterations >= 1


@pytest.mark.gpu
def test_ctx_trim_regrows(ctx):
    """gh_ctx_trim gives the context's grown buffers back; the next solves re-grow them and return the same numbers."""
    from gslam_amd import ba
    from gslam_amd.ba_synth import make_graph
    g = make_graph(160, 8000, n_obs_per_point=6, seed=5)
    p0, x0, s0, _ = ba.solve(ctx, g, ba.default_options(max_iterations=10))
    ctx.trim()
    p1, x1, s1, _ = ba.solve(ctx, g, ba.default_options(max_iterations=10))
    assert s0.iterations == s1.iterations and s0.final_cost == s1.final_cost
    assert np.array_equal(p0, p1) and np.array_equal(x0, x1)
    S = make_band(1000, 60, seed=8)
    b = np.ones(1000)
    xa, _ = ba.band_solve(ctx, S, b, 60)
    ctx.trim()
    xb, _ = ba.band_solve(ctx, S, b, 60)
    assert np.array_equal(xa, xb)


@pytest.mark.gpu
def test_ba_band_solver_with_masked_dofs_fixed_points_and_information(ctx):
    """The band path sees the same S as the dense one whatever the graph carries: fixed and partly fixed cameras inside the
    trajectory, fixed points, per-observation information matrices."""
    from gslam_amd.ba_synth import make_graph
    g = make_graph(200, 10000, n_obs_per_point=6, seed=11)
    rng = np.random.default_rng(11)
    dof = np.array(g["cam_dof"], dtype=np.int32).copy()
    dof[50] = 0          # fixed in the middle of the band
    dof[51] = 7          # translation only
    dof[120] = 56        # rotation only
    g["cam_dof"] = dof
    pf = np.ones(len(g["point_xyz"]), np.uint8)
    pf[rng.choice(len(pf), 300, replace=False)] = 0
    g["point_free"] = pf
    M = rng.standard_normal((len(g["obs_cam"]), 2, 2)) * 0.2
    g["obs_info"] = np.ascontiguousarray((M @ M.transpose(0, 2, 1) + np.eye(2)).reshape(-1, 4))
    pd, xd, sd, used_d = _solve_with(ctx, g, "dense", iters=25)
    pb, xb, sb, used_b = _solve_with(ctx, g, "band", iters=25)
    assert used_d[0] == "dense" and used_b[0] == "band"
    assert sd.iterations == sb.iterations
    n = sd.trace_len
    assert list(sd.trace_accepted[:n]) == list(sb.trace_accepted[:n])
    cd, cb = np.array(sd.trace_cost[:n]), np.array(sb.trace_cost[:n])
    assert np.abs(cd - cb).max() <= 1e-9 * np.abs(cd).max()
    assert np.abs(pd - pb).max() <= 1e-8 and np.abs(xd - xb).max() <= 1e-8
    assert np.array_equal(pb[50], np.asarray(g["cam_pose"])[50])            # the fixed camera did not move
    assert np.array_equal(xb[pf == 0], np.asarray(g["point_xyz"])[pf == 0])


@pytest.mark.gpu
@pytest.mark.parametrize("border", ["cameras", "points"])
@pytest.mark.parametrize("deterministic", [1, 0])
def test_ba_borders_with_masked_dofs_fixed_points_and_information(ctx, monkeypatch, border, deterministic):
    """Both kinds of arrowhead border (round 6) on a graph that carries everything the assembly has to respect: a fixed and partly
    fixed cameras (one of them an observer of a closure point), FIXED closure points and free ones, per-observation information
    matrices -- against the dense solver; with the deterministic assembly and with the atomics one (whose pair skip is separate code)."""
    from gslam_amd import ba
    from gslam_amd.ba_synth import make_graph
    monkeypatch.setenv("GSLAM_HIP_BA_POINT_BORDER", "1" if border == "points" else "0")
    g = make_graph(300, 20000, n_obs_per_point=6, seed=21, loop_closures=12)
    rng = np.random.default_rng(21)
    cp = np.asarray(g["closure_points"])
    oc, op = np.asarray(g["obs_cam"]), np.asarray(g["obs_point"])
    dof = np.array(g["cam_dof"], dtype=np.int32).copy()
    far = int(oc[op == cp[0]].max())
    dof[far] = 0                       # a fixed far observer of a closure point
    dof[int(oc[op == cp[1]].min())] = 7
    dof[150] = 56
    g["cam_dof"] = dof
    pf = np.ones(len(g["point_xyz"]), np.uint8)
    pf[rng.choice(len(pf), 400, replace=False)] = 0
    pf[cp[2]] = 0                      # a fixed closure point
    pf[cp[3]] = 1
    g["point_free"] = pf
    M = rng.standard_normal((len(oc), 2, 2)) * 0.2
    g["obs_info"] = np.ascontiguousarray((M @ M.transpose(0, 2, 1) + np.eye(2)).reshape(-1, 4))
    opts = lambda: ba.default_options(max_iterations=25, deterministic=deterministic)
    ctx.set_ba_solver("dense")
    try:
        pd, xd, sd, _ = ba.solve(ctx, g, opts())
        assert ctx.last_ba_solver()[0] == "dense"
    finally:
        ctx.set_ba_solver("auto")
    pa, xa, sa, _ = ba.solve(ctx, g, opts())
    assert ctx.last_ba_solver()[0] == "arrow" and (ctx.last_ba_border_points() == 12) == (border == "points")
    n = sd.trace_len
    assert sd.iterations == sa.iterations and list(sd.trace_accepted[:n]) == list(sa.trace_accepted[:n])
    cd, ca = np.array(sd.trace_cost[:n]), np.array(sa.trace_cost[:n])
    tol = 1e-9 if deterministic else 1e-7
    assert np.abs(cd - ca).max() <= tol * np.abs(cd).max()
    assert np.abs(pd - pa).max() <= 10 * tol and np.abs(xd - xa).max() <= 10 * tol
    assert np.array_equal(pa[far], np.asarray(g["cam_pose"])[far])
    assert np.array_equal(xa[pf == 0], np.asarray(g["point_xyz"])[pf == 0])
