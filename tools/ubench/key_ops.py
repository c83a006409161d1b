"""Micro-benchmark of the key arithmetic of gs2mesh_amd.parallel.canonical_keys (n keys on the GPU): per-variant median time,
device drained after each.  python tools/ubench/key_ops.py [n]"""
import statistics
import sys
import time

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3960
dev = torch.device("cuda:0")
B = 1 << 20
keys = torch.randint(-40, 40, (n, 3), dtype=torch.int32, device=dev)
mult = torch.tensor([1 << 42, 1 << 21, 1], dtype=torch.int64, device=dev)
shifts = torch.tensor([42, 21, 0], dtype=torch.int64, device=dev)


def pack_shift(k):
    k = k.to(torch.int64) + B
    return (k[:, 0] << 42) | (k[:, 1] << 21) | k[:, 2]


def pack_mulsum(k):
    return ((k + B).to(torch.int64) * mult).sum(dim=1)


def unpack_stack(u):
    return (torch.stack([(u >> 42) & 0x1FFFFF, (u >> 21) & 0x1FFFFF, u & 0x1FFFFF], dim=1) - B).to(torch.int32).contiguous()


def unpack_bcast(u):
    return (((u[:, None] >> shifts) & 0x1FFFFF) - B).to(torch.int32)


def unpack_div(u):
    # no tensor-tensor shift: floor division by constants through broadcasting multiplies is not exact for int64 -> use shifts by scalars on views
    out = torch.empty((u.shape[0], 3), dtype=torch.int64, device=u.device)
    torch.bitwise_right_shift(u, 42, out=out[:, 0])
    torch.bitwise_right_shift(u, 21, out=out[:, 1])
    out[:, 2] = u
    return ((out & 0x1FFFFF) - B).to(torch.int32)


u0 = pack_shift(keys)
assert torch.equal(u0, pack_mulsum(keys))
uu = torch.unique(u0, sorted=True)
assert torch.equal(unpack_stack(uu), unpack_bcast(uu)) and torch.equal(unpack_stack(uu), unpack_div(uu))
g = torch.zeros((1, 32770, 3), dtype=torch.int32, device=dev)
cases = {
    "pack_shift": lambda: pack_shift(keys), "pack_mulsum": lambda: pack_mulsum(keys),
    "unique": lambda: torch.unique(u0, sorted=True), "sort_only": lambda: torch.sort(u0),
    "unique_consecutive(sorted)": lambda: torch.unique_consecutive(uu),
    "unpack_stack": lambda: unpack_stack(uu), "unpack_bcast": lambda: unpack_bcast(uu), "unpack_div": lambda: unpack_div(uu),
    "head_cpu": lambda: g[:, 32768:, :].cpu(), "mask_select": lambda: keys[keys[:, 0] != 12345],
    "status_like_sync": lambda: torch.cuda.synchronize(),
    "host_tensor_to_dev": lambda: g[0, 32768:].copy_(torch.tensor([[1, 2, 3], [4, 5, 6]], dtype=torch.int32)),
}
for name, f in cases.items():
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"{name:32s} {1e6 * statistics.median(ts):8.1f} us")
