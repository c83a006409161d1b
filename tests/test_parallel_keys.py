"""Key arithmetic of the multi-GPU exchange (gs2mesh_amd/parallel.py): the packed form orders block indices lexicographically
over the whole +-2^20 key range of the volume (tsdf_common.h: GS2M_TSDF_KEY_BIAS), round-trips exactly, and the canonical list is
the sorted set union -- the properties every rank relies on to build the SAME list without talking to the others."""
import numpy as np
import torch

from gs2mesh_amd.parallel import _SENTINEL, _lex_unique, _pack_keys, _unpack_keys, exchange_window_from_cameras, shard_range

B = 1 << 20


def test_pack_is_exact_and_monotone_over_the_whole_key_range():
    rng = np.random.default_rng(11)
    k = rng.integers(-B, B, (5000, 3), dtype=np.int64)
    k[:8] = [[-B, -B, -B], [B - 1, B - 1, B - 1], [0, 0, 0], [-1, -1, -1], [-B, B - 1, 0], [B - 1, -B, -1], [0, 0, -B], [0, -B, 0]]
    t = torch.from_numpy(k.astype(np.int32))
    u = _pack_keys(t)
    assert u.dtype == torch.int64 and u.min() >= 0
    want = ((k[:, 0] + B) << 42) | ((k[:, 1] + B) << 21) | (k[:, 2] + B)
    assert np.array_equal(u.numpy(), want)
    back = _unpack_keys(u)
    assert back.dtype == torch.int32 and np.array_equal(back.numpy(), k)
    # ascending packed value == lexicographic (x, y, z) order
    order = np.lexsort((k[:, 2], k[:, 1], k[:, 0]))
    assert np.array_equal(np.sort(u.numpy()), want[order])
    # the sentinel (a block index nobody owns) packs above every real x
    s = _pack_keys(torch.tensor([[_SENTINEL, 0, 0]], dtype=torch.int32))
    assert int(s[0]) > int(_pack_keys(torch.tensor([[_SENTINEL - 1, B - 1, B - 1]], dtype=torch.int32))[0])


def test_canonical_list_is_the_sorted_set_union_whatever_the_order_of_the_parts():
    rng = np.random.default_rng(5)
    a = rng.integers(-30, 30, (400, 3)).astype(np.int32)
    b = np.concatenate([a[:150], rng.integers(-30, 30, (300, 3)).astype(np.int32)])
    want = np.unique(np.concatenate([a, b]), axis=0)          # numpy: lexicographically sorted unique rows
    for parts in ([a, b], [b, a], [b[::-1].copy(), a[::-1].copy()]):
        got = _lex_unique(torch.from_numpy(np.concatenate(parts)))
        assert got.dtype == torch.int32 and np.array_equal(got.numpy(), want)
    assert _lex_unique(torch.zeros((0, 3), dtype=torch.int32)).shape == (0, 3)
    one = _lex_unique(torch.tensor([[3, -2, 7]] * 4, dtype=torch.int32))
    assert one.tolist() == [[3, -2, 7]]


def test_shard_range_is_contiguous_balanced_and_complete():
    for n in (0, 1, 7, 16, 49, 300):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_exchange_window_from_cameras_covers_every_reachable_block():
    rng = np.random.default_rng(9)
    centres = rng.normal(size=(40, 3)) * 3.0 + np.array([120.0, -45.0, 7.0])        # a scene far from the origin (DTU-like units)
    voxel, max_depth = 0.02, 2.5
    lo, dim = exchange_window_from_cameras(centres, max_depth, voxel)
    unit = voxel * 16
    # any point within max_depth (+ truncation slack of half a block) of any camera falls into the window
    pts = centres[rng.integers(0, 40, 5000)] + rng.uniform(-1, 1, (5000, 3)) * max_depth
    b = np.floor(pts / unit).astype(np.int64)
    assert (b >= np.array(lo)).all() and (b < np.array(lo) + np.array(dim)).all()
    assert max(dim) < 2 * (6 * 3.0 + 2 * max_depth) / unit          # and it is not absurdly large
    # order of the cameras does not matter: every rank gets the same window
    assert exchange_window_from_cameras(centres[::-1], max_depth, voxel) == (lo, dim)
