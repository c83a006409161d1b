"""bench.py end to end on the GPU box (short jobs): the strong-scaling C4 job with the TSDF exchange forced at world size 1
(shard_range, the packed payload, the RCCL calls and exchange_halo's empty path at least run on one GPU),
and the --gpus contract."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, timeout=600):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_c4_strong_scaling_job_with_the_exchange_at_world_one():
    r = _bench(["--config", "C4", "--steps", "16", "--warmup", "2", "--scaling", "strong", "--always-collective", "--no-c3",
                "--no-cpu-baseline", "--no-parity", "--no-steady-state", "--min-repeats", "2", "--min-seconds", "0"])
    assert r.returncode == 0, r.stderr[-3000:]
    # ONE line on stdout, whatever the collective library prints (RCCL's version banner goes to C stdout: bench.py moves
    # descriptor 1 to stderr before the process group is created)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), lines[:8]
    line = lines[0]
    out = json.loads(line)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_C4_strong_world1.json"), "w") as fh:
        fh.write(line + "\n")
    assert out["n_gpus"] == 1 and out["scaling"] == "strong" and out["steps"] == 16
    red = out["tsdf"]["reduce"]
    assert red["payload"] == "packed" and red["frames_total"] == 16 and red["collectives"] >= 2 and red["always_collective"]
    assert red["bytes_per_rank"] == red["union_blocks"] * 4096 * 12 and red["halo_blocks_after"] == 0
    assert 0 < red["frac_of_timed_region"] < 1.0


def test_more_ranks_than_gpus_is_an_error():
    import torch
    n = torch.cuda.device_count()
    r = _bench(["--gpus", str(n + 1), "--steps", "2", "--warmup", "1"], timeout=120)
    assert r.returncode != 0 and "GPU(s) are visible" in r.stderr, (r.stdout[-300:], r.stderr[-500:])
