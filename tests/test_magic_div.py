"""gh_magic_div (host only): the multiplier orb_fast_cells uses instead of integer divisions must reproduce n / d for every n up
to the bound it was asked for -- checked exhaustively for small bounds, on the boundaries and at random for large ones -- and
must refuse (0) when no 32-bit constant can be guaranteed."""
import numpy as np


def test_magic_div_is_exact_or_refuses():
    from gslam_amd import hip
    hip.bind(strict=False)
    f = hip.lib.gh_magic_div
    rng = np.random.default_rng(1)
    cases = [(d, n_max) for d in range(2, 70) for n_max in (1, d - 1, d, d + 1, 1000, 4095)]
    cases += [(510, 510 * 1000), (30, 510), (17 * 9, 17 * 9 * 2000), (3, (1 << 30) - 1), (7, (1 << 30) - 1), (12345, (1 << 30) - 1),
              (65537, (1 << 30) - 1), ((1 << 20) + 7, (1 << 30) - 1)]
    cases += [(int(rng.integers(2, 1 << 16)), int(rng.integers(1, 1 << 30))) for _ in range(300)]
    refused = []
    for d, n_max in cases:
        m = int(f(d, n_max))
        if m == 0:
            refused.append((d, n_max))
            continue
        if n_max <= 4095:
            n = np.arange(n_max + 1, dtype=np.uint64)
        else:
            k = np.arange(0, n_max // d + 1, max(1, (n_max // d) // 20000), dtype=np.uint64) * np.uint64(d)
            n = np.unique(np.concatenate([k, k + np.uint64(d - 1), k[k > 0] - np.uint64(1), rng.integers(0, n_max + 1, 20000).astype(np.uint64),
                                          np.array([0, 1, n_max - 1, n_max], np.uint64)]))
            n = n[n <= n_max]
        assert np.array_equal((n * np.uint64(m)) >> np.uint64(32), n // np.uint64(d)), (d, n_max, m)
    assert f(0, 10) == 0 and f(1, 10) == 0
    # refusals only where the bound is far beyond anything a launch has (n_max * (m * d - 2^32) >= 2^32): never for small bounds
    assert all(n_max > (1 << 16) for d, n_max in refused) and len(refused) < len(cases)
    # the bench geometry: 510 tiles per 1080p frame, 1000 frames; 30 tile columns
    assert f(510, 510 * 1000) != 0 and f(30, 510) != 0
