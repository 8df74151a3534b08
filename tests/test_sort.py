"""Radix sort of the binning stage (include/gspl_hip.h §10) against numpy's stable sort: bit-exact keys AND values
(stability is part of the contract: the tile sort relies on it to keep the depth order inside every tile)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref_pairs(keys: np.ndarray, vals: np.ndarray, b0: int, b1: int):
    sel = (keys.astype(np.uint64) >> np.uint64(b0)) & np.uint64((1 << (b1 - b0)) - 1)
    order = np.argsort(sel, kind="stable")
    return keys[order], vals[order]


def _keys_u32(rng, n, kind):
    if kind == "uniform":
        return rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    if kind == "few":                      # long runs of equal keys: every tile publishes the same few digits
        return rng.choice(np.array([0, 1, 0x00FF00FF, 0x80000000, 0xFFFFFFFF], dtype=np.uint32), size=n)
    if kind == "depths":                   # positive floats of a narrow range + the "no tiles" sentinel, as bin_count sorts
        d = rng.uniform(2.7, 5.3, size=n).astype(np.float32).view(np.uint32)
        d[rng.random(n) < 0.05] = 0xFFFFFFFF
        return d
    if kind == "sorted":
        return np.sort(rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32))
    if kind == "reversed":
        return np.sort(rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32))[::-1].copy()
    raise KeyError(kind)


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 2047, 2048, 2049, 4096, 70_001, 1_000_000, 3_300_001])      # the last: several tiles per workgroup
@pytest.mark.parametrize("kind", ["uniform", "few", "depths"])
def test_pairs_u32_full_key(n, kind):
    import gspl_amd  # noqa: F401
    from gspl_amd import ops
    rng = np.random.default_rng(n * 7 + len(kind))
    keys = _keys_u32(rng, n, kind)
    vals = np.arange(n, dtype=np.uint32)
    rk, rv = _ref_pairs(keys, vals, 0, 32)
    k, v = ops.radix_sort_pairs(torch.from_numpy(keys.view(np.int32)).to(DEV), torch.from_numpy(vals.view(np.int32)).to(DEV))
    torch.cuda.synchronize()
    assert np.array_equal(k.cpu().numpy().view(np.uint32), rk)
    assert np.array_equal(v.cpu().numpy().view(np.uint32), rv)


@pytest.mark.parametrize("bits", [(0, 1), (0, 7), (3, 12), (8, 24), (5, 32), (31, 32), (0, 9)])
@pytest.mark.parametrize("kind", ["uniform", "sorted", "reversed"])
def test_pairs_u32_bit_ranges(bits, kind):
    import gspl_amd  # noqa: F401
    from gspl_amd import ops
    n = 300_007
    rng = np.random.default_rng(bits[0] * 100 + bits[1])
    keys = _keys_u32(rng, n, kind)
    vals = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    rk, rv = _ref_pairs(keys, vals, *bits)
    k, v = ops.radix_sort_pairs(torch.from_numpy(keys.view(np.int32)).to(DEV), torch.from_numpy(vals.view(np.int32)).to(DEV), *bits)
    assert np.array_equal(k.cpu().numpy().view(np.uint32), rk)
    assert np.array_equal(v.cpu().numpy().view(np.uint32), rv)


@pytest.mark.parametrize("n,tiles", [(0, 8160), (1, 8160), (2049, 1), (555_555, 8160), (3_000_001, 2500), (13_818_945, 8160), (400_000, 70_000)])
def test_keys_u64_tile_sort(n, tiles):
    """The tile sort: records (tile << 32 | rank) sorted on the tile bits only; the low word must keep its input order."""
    import gspl_amd  # noqa: F401
    from gspl_amd import ops
    rng = np.random.default_rng(n + tiles)
    tile = rng.integers(0, tiles, size=n, dtype=np.uint64)
    if n > 10:
        tile[: n // 3] = tile[0]            # a very long list for one tile
    rec = (tile << np.uint64(32)) | np.arange(n, dtype=np.uint64)
    bits = max(1, int(np.ceil(np.log2(tiles))))
    order = np.argsort(tile, kind="stable")
    out = ops.radix_sort_keys64(torch.from_numpy(rec.view(np.int64)).to(DEV), 32, 32 + bits)
    assert np.array_equal(out.cpu().numpy().view(np.uint64), rec[order])


def test_sort_is_repeatable_and_leaves_no_state():
    """Back-to-back sorts of different sizes through freshly allocated (dirty) workspaces."""
    import gspl_amd  # noqa: F401
    from gspl_amd import ops
    rng = np.random.default_rng(5)
    for n in (100_000, 5000, 100_000, 2_000_000, 17):
        keys = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
        vals = np.arange(n, dtype=np.uint32)
        rk, rv = _ref_pairs(keys, vals, 0, 32)
        junk = torch.full((1 << 22,), -1, dtype=torch.int32, device=DEV)      # poison what the allocator hands out next
        del junk
        k, v = ops.radix_sort_pairs(torch.from_numpy(keys.view(np.int32)).to(DEV), torch.from_numpy(vals.view(np.int32)).to(DEV))
        assert np.array_equal(k.cpu().numpy().view(np.uint32), rk) and np.array_equal(v.cpu().numpy().view(np.uint32), rv)
