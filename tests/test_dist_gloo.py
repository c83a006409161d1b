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


def single_process_reference():
    from backends import make
    from gs2mesh_amd.integration import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume
    from test_tsdf_parity import frames
    be = make("emu")
    frs, K = frames(5, 128, 96, 140.0)
    W, H, fx, fy, cx, cy = K
    vol = ScalableTSDFVolume(2.0 / 96, 0.1, max_blocks=2048, lib=be.lib)
    for d, c, E in frs:
        vol.integrate(RGBDImage(c, d), PinholeCameraIntrinsic(W, H, fx, fy, cx, cy), E)
    return vol.download()


@pytest.mark.parametrize("world,mode", [(2, "allreduce"), (2, "reduce_scatter"), (3, "allreduce")])
def test_sharded_fusion_equals_single_process(tmp_path, world, mode):
    for attempt in range(3):   # the rendezvous port is picked, released and re-bound by rank 0: retry if it was taken
        port = free_port()
        procs = []
        for rank in range(world):
            env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), OMP_NUM_THREADS="2")
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), mode],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in procs]
        if all(p.returncode == 0 for p in procs):
            break
        rendezvous = any("address already in use" in o.lower() or "connect" in o.lower() or "timed out" in o.lower() for o in outs)
        assert rendezvous and attempt < 2, "\n".join(o[-3000:] for o in outs)
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
