"""Multi-GPU sharding of the render -> fuse path (new: the reference is single-process, single-GPU).

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI on ROCm; ``gloo`` for
the CPU tests).  Stereo views are independent units, so rank r renders and integrates a contiguous
chunk of the views into its own block-sparse volume with NO data-path collective; the only exchange
is one sum-reduction of the TSDF accumulators at the end:

  1. all-gather the block keys, build the canonical (sorted, unique) union on every rank;
  2. ``gs2m_tsdf_pack`` the local volume onto that list in SUM form
     (wsum = tsdf*weight f32, weight f32, rgb_sum i32) -- zeros where a block is not allocated;
  3. RCCL ``all_reduce(SUM)`` (every rank ends with the whole volume) or ``reduce_scatter`` (rank r
     ends owning a contiguous 1/R of the blocks: half the bytes on the wire);
  4. ``gs2m_tsdf_unpack`` (tsdf = wsum/weight).

Counts and colour sums are integers, hence exact and order-independent; only ``tsdf`` carries fp32
reassociation error (<= ~1e-6) relative to integrating all views on one GPU.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous, balanced (+-1) chunk [lo, hi) of n_items for `rank` (contiguous so that
    stereo_warm chains stay intact inside a chunk)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def _lex_unique(keys: torch.Tensor) -> torch.Tensor:
    """Sorted unique rows of an int32 [n,3] key tensor (deterministic, identical on every rank)."""
    if keys.shape[0] == 0:
        return keys
    k = keys.to(torch.int64) + (1 << 20)
    packed = (k[:, 0] << 42) | (k[:, 1] << 21) | k[:, 2]
    u = torch.unique(packed, sorted=True)
    out = torch.stack([(u >> 42) & 0x1FFFFF, (u >> 21) & 0x1FFFFF, u & 0x1FFFFF], dim=1) - (1 << 20)
    return out.to(torch.int32).contiguous()


def canonical_keys(volume, group=None, always_collective: bool = False) -> torch.Tensor:
    """Union of the block keys of all ranks, canonical order, on the local device."""
    keys = volume.block_keys()
    if not torch.is_tensor(keys):
        keys = torch.from_numpy(np.ascontiguousarray(keys))
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (always_collective and dist.is_initialized()):
        return _lex_unique(keys)
    n_local = torch.tensor([keys.shape[0]], dtype=torch.int64, device=keys.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    n_max = int(max(int(c.item()) for c in counts))
    pad = torch.zeros((n_max, 3), dtype=torch.int32, device=keys.device)
    pad[: keys.shape[0]] = keys
    gathered = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(gathered, pad, group=group)
    allk = torch.cat([g[: int(c.item())] for g, c in zip(gathered, counts)], dim=0)
    return _lex_unique(allk)


def reduce_volume(volume, group=None, mode: str = "allreduce", always_collective: bool = False):
    """Sum-reduce the TSDF accumulators of all ranks into `volume`.

    mode "allreduce": every rank ends with the complete fused volume.
    mode "reduce_scatter": rank r ends with blocks [lo_r, hi_r) of the canonical list only.
    ``always_collective``: issue the collectives even at world size 1 (exercises the RCCL calls on one GPU).
    Returns dict(n_blocks_union, bytes_per_rank, keys) for reporting."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    keys = canonical_keys(volume, group, always_collective)
    collective = world > 1 or (always_collective and dist.is_initialized())
    n = int(keys.shape[0])
    dev = keys.device
    has_color = int(volume.color_type) == 1
    if mode == "reduce_scatter" and collective:
        n_pad = (n + world - 1) // world * world
    else:
        n_pad = n
    kpad = keys
    if n_pad != n:
        # pad with a key nobody owns (far outside any scene): packs to zeros, never unpacked
        filler = torch.full((n_pad - n, 3), (1 << 20) - 1, dtype=torch.int32, device=dev)
        kpad = torch.cat([keys, filler], dim=0).contiguous()
    wsum = torch.empty((n_pad, 4096), dtype=torch.float32, device=dev)
    weight = torch.empty((n_pad, 4096), dtype=torch.float32, device=dev)
    rgb = torch.empty((n_pad, 3, 4096), dtype=torch.int32, device=dev) if has_color else None
    if n_pad:
        volume.pack(kpad, wsum, weight, rgb)
    nbytes = n_pad * 4096 * (8 + (12 if has_color else 0))
    if collective and n_pad:
        if mode == "allreduce":
            dist.all_reduce(wsum, op=dist.ReduceOp.SUM, group=group)
            dist.all_reduce(weight, op=dist.ReduceOp.SUM, group=group)
            if has_color:
                dist.all_reduce(rgb, op=dist.ReduceOp.SUM, group=group)
            lo, hi = 0, n
        elif mode == "reduce_scatter":
            per = n_pad // world
            o_ws = torch.empty((per, 4096), dtype=torch.float32, device=dev)
            o_w = torch.empty((per, 4096), dtype=torch.float32, device=dev)
            dist.reduce_scatter_tensor(o_ws, wsum, op=dist.ReduceOp.SUM, group=group)
            dist.reduce_scatter_tensor(o_w, weight, op=dist.ReduceOp.SUM, group=group)
            if has_color:
                o_c = torch.empty((per, 3, 4096), dtype=torch.int32, device=dev)
                dist.reduce_scatter_tensor(o_c, rgb, op=dist.ReduceOp.SUM, group=group)
                rgb = o_c
            wsum, weight = o_ws, o_w
            lo, hi = rank * per, min(n, (rank + 1) * per)
            kpad = kpad[lo: lo + per].contiguous()
        else:
            raise ValueError(mode)
    else:
        lo, hi = 0, n
    # replace the local state by the reduced blocks
    volume.status()
    volume.reset()
    cnt = max(0, hi - lo)
    if cnt:
        volume.unpack(kpad[:cnt].contiguous(), wsum[:cnt].contiguous(), weight[:cnt].contiguous(),
                      rgb[:cnt].contiguous() if has_color else None)
    volume.status()
    return dict(n_blocks_union=n, bytes_per_rank=nbytes, keys=keys, owned=(lo, hi))
