"""Multi-GPU sharding of the render -> fuse path (new: the reference is single-process, single-GPU).

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI on ROCm; ``gloo`` for
the CPU tests).  Stereo views are independent units, so rank r renders and integrates a contiguous
chunk of the views into its own block-sparse volume with NO data-path collective; the only exchange
is one sum-reduction of the TSDF accumulators at the end (``reduce_volume``):

  1. ONE fixed-size ``all_gather`` of the block keys (``max_blocks`` rows per rank, sentinel-padded, plus one
     row carrying the rank's block count and overflow flags -- no size handshake, no per-rank ``.item()``);
     every rank builds the same canonical (sorted, unique) union; an overflow on ANY rank raises on EVERY
     rank after the collective (nobody is left waiting in one);
  2. ``gs2m_tsdf_pack_sum`` writes the local accumulators of the union blocks in SUM form into ONE fp32
     buffer ``[n, 5, 4096]`` (wsum = tsdf*weight, weight, sum r, sum g, sum b; the counts and colour sums are
     integers < 2^24, exact in fp32 and independent of the reduction order);
  3. ONE RCCL collective over that buffer: ``reduce_scatter`` (default: rank r ends owning a contiguous 1/R of
     the canonical block list -- (R-1)/R of the bytes of an all-reduce on the wire, and xGMI is per-link
     bound) or ``all_reduce`` (every rank ends with the complete volume);
  4. ``gs2m_tsdf_unpack_sum`` (tsdf = wsum/weight) of the owned blocks.

After a reduce-scatter every rank extracts ITS part of the mesh (owner-side finalisation): ``exchange_halo``
fetches the +1 neighbour blocks that belong to other ranks (one ``all_to_all``, sizes derived from the
canonical list on every rank alike -- no handshake) and marks them neighbour-only, so that the cubes on the
boundary of a rank's part see all 8 corners and every cube is produced by exactly one rank.

Only ``tsdf`` carries fp32 reassociation error (<= ~1e-6) relative to integrating all views on one GPU.
"""
from __future__ import annotations

import time

import numpy as np
import torch
import torch.distributed as dist

_SENTINEL = (1 << 20) - 1          # block index nobody owns (outside the +-2^20 key range of the volume)
_OVERFLOW_TEXT = ((1, "block pool exhausted (raise max_blocks)"), (2, "hash table full"),
                  (4, "block index out of the +-2^20 range"))


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous, balanced (+-1) chunk [lo, hi) of n_items for `rank` (contiguous so that
    stereo_warm chains stay intact inside a chunk)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def _pack_keys(keys: torch.Tensor) -> torch.Tensor:
    k = keys.to(torch.int64) + (1 << 20)
    return (k[:, 0] << 42) | (k[:, 1] << 21) | k[:, 2]


def _unpack_keys(u: torch.Tensor) -> torch.Tensor:
    out = torch.stack([(u >> 42) & 0x1FFFFF, (u >> 21) & 0x1FFFFF, u & 0x1FFFFF], dim=1) - (1 << 20)
    return out.to(torch.int32).contiguous()


def _lex_unique(keys: torch.Tensor) -> torch.Tensor:
    """Sorted unique rows of an int32 [n,3] key tensor (deterministic, identical on every rank)."""
    if keys.shape[0] == 0:
        return keys
    return _unpack_keys(torch.unique(_pack_keys(keys), sorted=True))


def _as_tensor(keys):
    return keys if torch.is_tensor(keys) else torch.from_numpy(np.ascontiguousarray(keys))


def canonical_keys(volume, group=None, always_collective: bool = False):
    """Union of the block keys of all ranks in canonical order, on the local device, + the OR of the ranks'
    overflow flags.  One fixed-size all_gather; one host read (of the gathered header rows)."""
    keys = _as_tensor(volume.block_keys(raise_on_overflow=False))
    _, _, ov = volume.status(raise_on_overflow=False)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    collective = world > 1 or (always_collective and dist.is_initialized())
    if not collective:
        return _lex_unique(keys), int(ov), 0
    K = int(volume.max_blocks)
    n_local = int(keys.shape[0])
    buf = torch.full((K + 1, 3), _SENTINEL, dtype=torch.int32, device=keys.device)
    buf[:n_local] = keys
    buf[K, 0] = n_local
    buf[K, 1] = int(ov)
    buf[K, 2] = K
    gathered = torch.empty((world * (K + 1), 3), dtype=torch.int32, device=keys.device)
    dist.all_gather_into_tensor(gathered, buf, group=group)
    g = gathered.view(world, K + 1, 3)
    head = g[:, K, :].cpu()                                   # the one host read of the exchange
    if int(head[:, 2].min()) != K or int(head[:, 2].max()) != K:
        raise RuntimeError("reduce_volume: every rank must create its volume with the same max_blocks")
    ov_any = 0
    for f in head[:, 1].tolist():
        ov_any |= int(f)
    body = g[:, :K, :].reshape(-1, 3)
    valid = body[:, 0] != _SENTINEL
    return _lex_unique(body[valid]), ov_any, 1


def reduce_volume(volume, group=None, mode: str = "reduce_scatter", always_collective: bool = False):
    """Sum-reduce the TSDF accumulators of all ranks into `volume`.

    mode "reduce_scatter": rank r ends with blocks [lo_r, hi_r) of the canonical list only (use `exchange_halo`
    before extracting its part of the mesh).  mode "allreduce": every rank ends with the complete fused volume.
    ``always_collective``: issue the collectives even at world size 1 (exercises the RCCL calls on one GPU).
    The caller must have drained the streams that integrate into `volume` (`RenderFusePipeline.drain`).
    Returns dict(n_blocks_union, bytes_per_rank, keys, owned, collectives, seconds)."""
    t0 = time.perf_counter()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    keys, ov_any, n_coll = canonical_keys(volume, group, always_collective)
    if ov_any:
        # every rank sees the same flags after the key exchange: all of them raise, nobody waits in a collective
        what = [n for b, n in _OVERFLOW_TEXT if ov_any & b]
        raise RuntimeError("TSDF volume overflow on at least one rank: " + ", ".join(what))
    collective = world > 1 or (always_collective and dist.is_initialized())
    n = int(keys.shape[0])
    dev = keys.device
    scatter = mode == "reduce_scatter" and collective
    if mode not in ("reduce_scatter", "allreduce"):
        raise ValueError(mode)
    n_pad = (n + world - 1) // world * world if scatter else n
    kpad = keys
    if n_pad != n:
        # pad with a key nobody owns (far outside any scene): packs to zeros, never unpacked
        filler = torch.full((n_pad - n, 3), _SENTINEL, dtype=torch.int32, device=dev)
        kpad = torch.cat([keys, filler], dim=0).contiguous()
    buf = torch.empty((n_pad, 5, 4096), dtype=torch.float32, device=dev)
    if n_pad:
        volume.pack_sum(kpad, buf)
    nbytes = n_pad * 5 * 4096 * 4
    lo, hi = 0, n
    if collective and n_pad:
        if scatter:
            per = n_pad // world
            out = torch.empty((per, 5, 4096), dtype=torch.float32, device=dev)
            dist.reduce_scatter_tensor(out, buf, op=dist.ReduceOp.SUM, group=group)
            buf = out
            lo, hi = min(n, rank * per), min(n, (rank + 1) * per)
            kpad = kpad[rank * per: (rank + 1) * per]
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        n_coll += 1
    # replace the local state by the reduced blocks (reset only clears the slots in use)
    volume.reset()
    cnt = max(0, hi - lo)
    if cnt:
        volume.unpack_sum(kpad[:cnt].contiguous(), buf[:cnt])
    volume.status()
    return dict(n_blocks_union=n, bytes_per_rank=nbytes, keys=keys, owned=(lo, hi), collectives=n_coll, mode=mode,
                per=(n_pad // world if scatter else n), seconds=time.perf_counter() - t0)


def _neighbour_index(packed_sorted: torch.Tensor, keys: torch.Tensor, d):
    """Index in the canonical list of block key + d for every key (-1 where it is not allocated anywhere)."""
    q = _pack_keys(keys + torch.tensor(d, dtype=torch.int32, device=keys.device))
    pos = torch.searchsorted(packed_sorted, q)
    pos = torch.clamp(pos, max=packed_sorted.shape[0] - 1)
    return torch.where(packed_sorted[pos] == q, pos, torch.full_like(pos, -1))


def exchange_halo(volume, info, group=None):
    """Owner-side finalisation after ``reduce_volume(mode="reduce_scatter")``: fetch the blocks of other ranks that
    the cubes of this rank's blocks reach (+1 neighbours in x / y / z: marching cubes looks one voxel ahead) and mark
    them neighbour-only.  Which rank needs which block follows from the canonical key list and the ownership split,
    identical on every rank: ONE all_to_all, no handshake.  Returns the number of halo blocks received."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    keys = info["keys"]
    n = int(keys.shape[0])
    if world == 1 or n == 0 or info.get("mode") != "reduce_scatter":
        return 0
    dev = keys.device
    per = int(info["per"])
    packed = _pack_keys(keys)                                  # canonical order = ascending packed key
    owner = torch.arange(n, device=dev) // per
    # (needing rank p, provider block j): block i owned by p has a +1 neighbour j owned by somebody else
    need = torch.zeros((world, n), dtype=torch.bool, device=dev)
    for d in [(a, b, c) for a in (0, 1) for b in (0, 1) for c in (0, 1) if (a, b, c) != (0, 0, 0)]:
        j = _neighbour_index(packed, keys, d)
        ok = (j >= 0)
        jj = torch.where(ok, j, torch.zeros_like(j))
        ok = ok & (owner[jj] != owner)
        need[owner[ok], jj[ok]] = True
    mine = owner == rank
    send_idx = [torch.nonzero(need[p] & mine, as_tuple=False).flatten() for p in range(world)]   # my blocks rank p needs
    recv_idx = [torch.nonzero(need[rank] & (owner == p), as_tuple=False).flatten() for p in range(world)]
    send_counts = [int(x.numel()) for x in send_idx]
    recv_counts = [int(x.numel()) for x in recv_idx]
    s_all = torch.cat(send_idx) if sum(send_counts) else torch.zeros(0, dtype=torch.long, device=dev)
    r_all = torch.cat(recv_idx) if sum(recv_counts) else torch.zeros(0, dtype=torch.long, device=dev)
    sbuf = torch.empty((max(int(s_all.numel()), 1), 5, 4096), dtype=torch.float32, device=dev)
    if s_all.numel():
        volume.pack_sum(keys[s_all].contiguous(), sbuf[: s_all.numel()])
    rbuf = torch.empty((max(int(r_all.numel()), 1), 5, 4096), dtype=torch.float32, device=dev)
    dist.all_to_all_single(rbuf[: r_all.numel()].reshape(-1), sbuf[: s_all.numel()].reshape(-1),
                           output_split_sizes=[c * 5 * 4096 for c in recv_counts],
                           input_split_sizes=[c * 5 * 4096 for c in send_counts], group=group)
    if r_all.numel():
        volume.unpack_sum(keys[r_all].contiguous(), rbuf[: r_all.numel()], halo=True)
    volume.status()
    return int(r_all.numel())
