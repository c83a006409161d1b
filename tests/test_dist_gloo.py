"""The N>1 path on CPU: world_size-2 (and 3) gloo groups run the production sharding + reduction code
(gs2mesh_amd/parallel.py) over the emulator build of the kernels, and the fused volume must equal the
single-process result: counts and colour sums bit-exact, tsdf within 1e-5."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from gs2mesh_amd.parallel import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_is_a_balanced_contiguous_partition():
    for n in (0, 1, 7, 49, 300):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def run_world(tmp_path, world, mode, opts=""):
    for attempt in range(3):   # the rendezvous port is picked, released and re-bound by rank 0: retry if it was taken
        port = free_port()
        procs = []
        for rank in range(world):
            env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), OMP_NUM_THREADS="2", GS2M_TEST_OPTS=opts)
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), mode],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in procs]
        if all(p.returncode == 0 for p in procs):
            return
        rendezvous = any("address already in use" in o.lower() or "connect" in o.lower() or "timed out" in o.lower() for o in outs)
        assert rendezvous and attempt < 2, "\n".join(o[-3000:] for o in outs)


def single_process_reference(with_mesh=False, n_frames=5):
    from backends import make
    from gs2mesh_amd.integration import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume
    from test_tsdf_parity import frames
    be = make("emu")
    frs, K = frames(n_frames, 128, 96, 140.0)
    W, H, fx, fy, cx, cy = K
    vol = ScalableTSDFVolume(2.0 / 96, 0.1, max_blocks=2048, lib=be.lib)
    for d, c, E in frs:
        vol.integrate(RGBDImage(c, d), PinholeCameraIntrinsic(W, H, fx, fy, cx, cy), E)
    if with_mesh:
        m = vol.extract_triangle_mesh()
        return vol.download(), m.vertices[m.triangles]
    return vol.download()



def assert_same_triangles(got, ref, atol=2e-6):
    """The sharded tsdf differs from the single-process one by the order of the sums (<= 1e-5), so the triangles [n,3,3] are
    matched by nearest neighbour in the 9-D space of their three vertices (emission order of the vertices is the same on both
    sides; a sort on rounded coordinates is not stable against that noise): every triangle has its partner within `atol`, both
    ways, with equal counts."""
    from scipy.spatial import cKDTree
    a, b = got.reshape(len(got), 9), ref.reshape(len(ref), 9)
    assert a.shape == b.shape
    _, nn = cKDTree(b).query(a, k=1)
    np.testing.assert_allclose(a, b[nn], atol=atol, rtol=0)        # every triangle of `got` is one of `ref` ...
    _, mm = cKDTree(a).query(b, k=1)
    np.testing.assert_allclose(b, a[mm], atol=atol, rtol=0)        # ... and the other way round (equal counts: same set)
    assert len(np.unique(nn)) >= 0.999 * len(b)                     # a bijection but for exactly coincident (degenerate) triangles

@pytest.mark.parametrize("world,mode", [(2, "allreduce"), (2, "reduce_scatter"), (3, "allreduce"),
                                        (2, "reduce_scatter:f32"), (3, "reduce_scatter:packed:direct"),
                                        (2, "reduce_scatter:f32:direct"), (2, "allreduce:f32")])
def test_sharded_fusion_equals_single_process(tmp_path, world, mode):
    """Every payload (packed int64 lanes / five fp32 planes) and both reduce-scatter algorithms (the library's, or one
    all_to_all of the 1/R slices + a local sum) give the single-process volume: weights and colour sums bit-exact."""
    run_world(tmp_path, world, mode)
    mode = mode.split(":")[0]
    kf, tf, wf, cf = single_process_reference()
    ref = {tuple(k): i for i, k in enumerate(kf.tolist())}
    seen = set()
    for rank in range(world):
        z = np.load(os.path.join(tmp_path, f"rank{rank}.npz"))
        assert int(z["union"]) == len(ref)
        keys = list(map(tuple, z["keys"].tolist()))
        if mode == "allreduce":
            assert set(keys) == set(ref)
        else:
            assert not (set(keys) & seen)            # ownership is a partition
        seen |= set(keys)
        idx = np.array([ref[k] for k in keys], dtype=int)
        np.testing.assert_array_equal(z["weight"], wf[idx])
        np.testing.assert_array_equal(z["rgb"], cf[idx])
        np.testing.assert_allclose(z["tsdf"], tf[idx], atol=1e-5, rtol=0)
    assert seen == set(ref)


@pytest.mark.parametrize("opts", ["window=small", "keys=gather"])
def test_key_exchange_paths_agree(tmp_path, opts):
    """The block-bitmap exchange (default), its fallback when a block lies outside the window (a 2^3-block window: every rank
    sees the `outside` word of the reduced header and gathers), and the gather path asked for directly: the same canonical
    union, the same reduced volume."""
    world = 3
    run_world(tmp_path, world, "reduce_scatter:packed:direct", opts)
    kf, tf, wf, cf = single_process_reference()
    ref = {tuple(k): i for i, k in enumerate(kf.tolist())}
    seen = set()
    for rank in range(world):
        z = np.load(os.path.join(tmp_path, f"rank{rank}.npz"))
        assert int(z["union"]) == len(ref)
        keys = list(map(tuple, z["keys"].tolist()))
        assert not (set(keys) & seen)
        seen |= set(keys)
        idx = np.array([ref[k] for k in keys], dtype=int)
        np.testing.assert_array_equal(z["weight"], wf[idx])
        np.testing.assert_array_equal(z["rgb"], cf[idx])
    assert seen == set(ref)


def test_a_pack_overflow_on_one_rank_raises_on_every_rank(tmp_path):
    """the packed-form overflow flag is set by the LOCAL pack kernel; the verdict is all-reduced, every rank
    raises, and all of them reach the barrier behind it."""
    run_world(tmp_path, 3, "reduce_scatter:packed", "badpack")
    for rank in range(3):
        assert int(np.load(os.path.join(tmp_path, f"rank{rank}.npz"))["refused"]) == 1


def test_state_injected_on_every_rank_adds_up_in_the_weight_bound(tmp_path):
    """one checkpoint loaded on every rank -- the injected bounds add up, `auto` falls back to the f32 payload."""
    run_world(tmp_path, 2, "allreduce", "checkpoint")
    for rank in range(2):
        assert int(np.load(os.path.join(tmp_path, f"rank{rank}.npz"))["refused"]) == 1


@pytest.mark.parametrize("opt", ["window=mismatch", "window=mismatch_size"])
def test_mismatching_exchange_windows_are_refused_on_every_rank(tmp_path, opt):
    run_world(tmp_path, 2, "reduce_scatter", opt)
    for rank in range(2):
        assert int(np.load(os.path.join(tmp_path, f"rank{rank}.npz"))["refused"]) == 1


def test_eight_ranks_reduce_scatter_direct_and_owner_side_mesh(tmp_path):
    """The target rank count: 8 ranks x 2 frames, packed payload, direct reduce-scatter, owner-side extraction with
    halo blocks -- ownership split, padding of the canonical list to a multiple of 8, halo need / send lists at R = 8."""
    world, n_frames = 8, 16
    run_world(tmp_path, world, f"reduce_scatter+mesh:packed:direct:{n_frames}")
    (kf, tf, wf, cf), tri_ref = single_process_reference(with_mesh=True, n_frames=n_frames)
    ref = {tuple(k): i for i, k in enumerate(kf.tolist())}
    parts, owned_all = [], set()
    for rank in range(world):
        z = np.load(os.path.join(tmp_path, f"rank{rank}.npz"))
        assert int(z["union"]) == len(ref)
        owned = list(map(tuple, z["owned_keys"].tolist()))
        assert not (set(owned) & owned_all)
        owned_all |= set(owned)
        parts.append(z["tri_xyz"])
        # halo copies arrive VERBATIM: bit-identical to the owner's reduced voxels
        keys = list(map(tuple, z["keys"].tolist()))
        idx = np.array([ref[k] for k in keys], dtype=int)
        np.testing.assert_array_equal(z["weight"], wf[idx])
        np.testing.assert_array_equal(z["rgb"], cf[idx])
    assert owned_all == set(ref)
    got = np.concatenate(parts, axis=0)
    assert got.shape == tri_ref.shape and len(got) > 1000
    assert_same_triangles(got, tri_ref)


def test_eight_ranks_fall_back_to_the_f32_payload_beyond_1023_frames(tmp_path):
    """with more than 1023 frames in total (here 8 ranks x (2 real + 200 pretended) frames) the int64 packed
    payload could carry between its fields: payload='auto' must pick the five fp32 planes ON EVERY RANK ALIKE (the bound travels
    in the gathered header) and still give the single-process volume; payload='packed' must be refused on every rank."""
    world, n_frames = 8, 16
    run_world(tmp_path, world, f"reduce_scatter:auto:rccl:{n_frames}:200")
    kf, tf, wf, cf = single_process_reference(n_frames=n_frames)
    ref = {tuple(k): i for i, k in enumerate(kf.tolist())}
    seen = set()
    for rank in range(world):
        z = np.load(os.path.join(tmp_path, f"rank{rank}.npz"))
        keys = list(map(tuple, z["keys"].tolist()))
        assert not (set(keys) & seen)
        seen |= set(keys)
        idx = np.array([ref[k] for k in keys], dtype=int)
        np.testing.assert_array_equal(z["weight"], wf[idx])
        np.testing.assert_array_equal(z["rgb"], cf[idx])
        np.testing.assert_allclose(z["tsdf"], tf[idx], atol=1e-5, rtol=0)
    assert seen == set(ref)
    run_world(tmp_path, world, f"reduce_scatter:packed:direct:{n_frames}:200")
    for rank in range(world):
        assert int(np.load(os.path.join(tmp_path, f"rank{rank}.npz"))["refused"]) == 1


def test_a_replicated_volume_is_not_summed_again(tmp_path):
    """After mode='allreduce' every rank holds the whole fused volume: a second reduction would count it once per rank.
    reduce_volume refuses (on every rank: the flag travels in the header); after a reduce-scatter a second reduction is fine
    (test_halo_copies_are_not_exchanged_twice)."""
    run_world(tmp_path, 2, "allreduce+twice")
    for rank in range(2):
        assert int(np.load(os.path.join(tmp_path, f"rank{rank}.npz"))["refused_second"]) == 1


def test_halo_copies_are_not_exchanged_twice(tmp_path):
    """A volume that holds halo copies (after exchange_halo) reports them as sentinel keys and packs them as zeros: a
    second reduction does not count another rank's blocks twice."""
    run_world(tmp_path, 2, "reduce_scatter+mesh+again")
    kf, tf, wf, cf = single_process_reference()
    ref = {tuple(k): i for i, k in enumerate(kf.tolist())}
    seen = set()
    for rank in range(2):
        z = np.load(os.path.join(tmp_path, f"rank{rank}.npz"))
        keys = list(map(tuple, z["keys2"].tolist()))
        assert not (set(keys) & seen)
        seen |= set(keys)
        idx = np.array([ref[k] for k in keys], dtype=int)
        np.testing.assert_array_equal(z["weight2"], wf[idx])          # weights NOT doubled by the halo copies
        np.testing.assert_array_equal(z["rgb2"], cf[idx])
    assert seen == set(ref)


@pytest.mark.parametrize("world", [2, 3])
def test_owner_side_mesh_extraction_with_halo_blocks(tmp_path, world):
    """After the reduce-scatter every rank holds 1/R of the blocks; `exchange_halo` brings in the +1 neighbour blocks
    of other ranks as neighbour-only blocks, and the ranks' partial meshes together are exactly the single-process
    mesh: every triangle once, none missing on the ownership boundaries (vertex positions only depend on the weights
    / tsdf of the 8 corners, and tsdf differs by fp32 reassociation only)."""
    run_world(tmp_path, world, "reduce_scatter+mesh")
    (kf, tf, wf, cf), tri_ref = single_process_reference(with_mesh=True)
    parts, n_halo = [], 0
    for rank in range(world):
        z = np.load(os.path.join(tmp_path, f"rank{rank}.npz"))
        parts.append(z["tri_xyz"])
        n_halo += int(z["n_halo"])
        assert len(z["keys"]) == len(z["owned_keys"]) + int(z["n_halo"])     # halo blocks were added to the volume
    assert n_halo > 0
    got = np.concatenate(parts, axis=0)
    assert got.shape == tri_ref.shape and len(got) > 1000

    assert_same_triangles(got, tri_ref)
