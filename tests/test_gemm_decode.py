"""CPU: the work-item decode of the persistent GEMM kernels (csrc/gemm_impl.h `div_by` / `ring_item`, host side `set_tiles` in
csrc/gemm.hip) replaces integer divisions by a multiply-high with a host-computed reciprocal and ONE correction step.  This is
a restatement of that arithmetic in numpy, checked against true division over the whole admitted range (work items < 2^30,
divisors from 1), plus the tile walk itself against a plain-Python enumeration."""
import numpy as np


def inv_of(d):
    return 0xFFFFFFFF if d <= 1 else (1 << 32) // d


def div_by(n, d, inv):
    q = (n.astype(np.uint64) * np.uint64(inv)) >> np.uint64(32)
    q = q.astype(np.int64)
    return np.where(n - q * d >= d, q + 1, q)


def test_reciprocal_division_is_exact():
    rng = np.random.default_rng(0)
    ds = [1, 2, 3, 4, 5, 7, 12, 16, 48, 64, 82, 255, 256, 257, 1312, 4096, 65535, 65536, 1 << 20, (1 << 20) + 1, (1 << 30) - 1]
    ds += [int(x) for x in rng.integers(1, 1 << 22, size=200)]
    for d in ds:
        inv = inv_of(d)
        n = np.concatenate([np.arange(0, min(4 * d + 3, 1 << 16)), rng.integers(0, 1 << 30, size=4000),
                            np.array([d - 1, d, d + 1, 2 * d - 1, 2 * d, (1 << 30) - 1, (1 << 30) - d])]).astype(np.int64)
        n = n[(n >= 0) & (n < (1 << 30))]
        assert np.array_equal(div_by(n, d, inv), n // d), d


def ring_item(idx, tiles_m, tiles_n, gh):
    """(tm, tn) of work item idx of one K slice, as csrc/gemm_impl.h walks them: groups of `gh` tile rows, column by column."""
    per_panel = gh * tiles_n
    panel = int(div_by(np.array([idx]), per_panel, inv_of(per_panel))[0])
    r = idx - panel * per_panel
    left = tiles_m - panel * gh
    g = min(left, gh)
    tn = (r // gh) if g == gh else (r // g)
    tm = panel * gh + (r - tn * g)
    return tm, tn


def test_tile_walk_covers_every_tile_once():
    for tiles_m, tiles_n, gh in [(82, 16, 4), (82, 4, 4), (345, 12, 4), (1, 1, 4), (3, 5, 4), (7, 3, 8), (4, 1, 4), (359, 16, 4)]:
        seen = set()
        for idx in range(tiles_m * tiles_n):
            tm, tn = ring_item(idx, tiles_m, tiles_n, gh)
            assert 0 <= tm < tiles_m and 0 <= tn < tiles_n, (tiles_m, tiles_n, gh, idx, tm, tn)
            seen.add((tm, tn))
        assert len(seen) == tiles_m * tiles_n, (tiles_m, tiles_n, gh)
