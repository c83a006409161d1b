"""Multi-GPU sharding of the render -> fuse path (new: the reference is single-process, single-GPU).

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI on ROCm; ``gloo`` for
the CPU tests).  Stereo views are independent units, so rank r renders and integrates a contiguous
chunk of the views into its own block-sparse volume with NO data-path collective; the only exchange
is one sum-reduction of the TSDF accumulators at the end (``reduce_volume``):

  1. ONE fixed-size ``all_gather`` of the block keys (``max_blocks`` rows per rank, the first n of them valid, plus two
     header rows carrying the rank's block count, overflow flags and frame counts -- no size handshake, no per-rank
     ``.item()``; the lists are cut out of the gathered buffer by the header's counts);
     every rank builds the same canonical (sorted, unique) union; an overflow on ANY rank raises on EVERY
     rank after the collective (nobody is left waiting in one);
  2. ``gs2m_tsdf_pack`` writes the local accumulators of the union blocks in SUM form into persistent, grow-only
     buffers owned by the volume.  Payload "packed" (chosen when all ranks together integrated <= 1023 frames -- the
     frame counts travel in the header of step 1, so every rank decides alike): wsum = tsdf*weight as fp32 + ONE int64
     per voxel holding weight | sum r << 10 | sum g << 28 | sum b << 46 -- 12 bytes per voxel, integer fields exact under
     an integer SUM.  Payload "f32" (any frame count): five fp32 planes (wsum, weight, sum r, sum g, sum b; counts and
     colour sums are integers < 2^24, exact in fp32) -- 20 bytes per voxel;
  3. the sum: ``reduce_scatter`` (default: rank r ends owning a contiguous 1/R of the canonical block list --
     (R-1)/R of the bytes of an all-reduce on the wire, and xGMI is per-link bound), as RCCL's
     reduce-scatter or ("direct") as ONE all_to_all of the 1/R slices + a local sum (every pair of GPUs has its own
     xGMI link: each slice crosses one link once, no ring), or ``all_reduce`` (every rank ends with the whole volume);
  4. ``gs2m_tsdf_unpack`` (tsdf = wsum/weight) of the owned blocks.

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

_MAP_HEADER = 32                 # = _lib.TSDF_MAP_HEADER_BYTES (include/gs2mesh_amd.h)
_SENTINEL = (1 << 20) - 1          # block index nobody owns (outside the +-2^20 key range of the volume)
_OVERFLOW_TEXT = ((1, "block pool exhausted (raise max_blocks)"), (2, "hash table full"),
                  (4, "block index out of the +-2^20 range"),
                  (8, "a voxel did not fit the packed exchange form (weight > 1023 or colour sum >= 2^18)"))


def _via_host(group, *tensors) -> bool:
    """gloo takes device tensors for a few collectives only (all_reduce, broadcast): every other one -- and, to keep one code
    path, all of them -- is staged through host memory when the group is a gloo group and the buffers live on a GPU.  That is
    the REHEARSAL configuration (`bench.py --rehearsal`, tests/test_bench_gpu.py): the whole N > 1 control flow with the ranks
    sharing one GPU.  Under RCCL (backend "nccl") nothing is staged."""
    return dist.get_backend(group) == "gloo" and any(t.is_cuda for t in tensors)


def _all_reduce(t, op, group):
    if _via_host(group, t):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)


def _reduce_scatter_tensor(out, inp, group):
    if _via_host(group, out, inp):
        ho, hi = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
        dist.reduce_scatter_tensor(ho, hi, op=dist.ReduceOp.SUM, group=group)
        out.copy_(ho)
    else:
        dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=group)


def _all_to_all_single(out, inp, group, **kw):
    if _via_host(group, out, inp):
        ho, hi = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
        dist.all_to_all_single(ho, hi, group=group, **kw)
        out.copy_(ho)
    else:
        dist.all_to_all_single(out, inp, group=group, **kw)


def _all_gather_into_tensor(out, inp, group):
    if _via_host(group, out, inp):
        ho, hi = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
        dist.all_gather_into_tensor(ho, hi, group=group)
        out.copy_(ho)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def _mark(marks, name):
    """profiling aid (tools/reduce_breakdown.py): phase boundary, device drained; nothing happens without a list"""
    if marks is not None:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        marks.append((name, time.perf_counter()))


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous, balanced (+-1) chunk [lo, hi) of n_items for `rank` (contiguous so that
    stereo_warm chains stay intact inside a chunk)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


_CONST = {}      # (name, device) -> small constant tensors of the key arithmetic (created once: no host -> device copy per call)


def _const(name, values, device):
    t = _CONST.get((name, str(device)))
    if t is None:
        t = _CONST[(name, str(device))] = torch.tensor(values, dtype=torch.int64, device=device)
    return t


def _pack_keys(keys: torch.Tensor) -> torch.Tensor:
    """[n,3] int32 block indices -> one int64 per block (x | y | z biased to 21 bits each: ascending packed value =
    lexicographic order).  Four launches (bias, widen, scale, row sum)."""
    mult = _const("mult", [1 << 42, 1 << 21, 1], keys.device)
    return ((keys + (1 << 20)).to(torch.int64) * mult).sum(dim=1)


def _unpack_keys(u: torch.Tensor) -> torch.Tensor:
    shifts = _const("shifts", [42, 21, 0], u.device)
    return (((u[:, None] >> shifts) & 0x1FFFFF) - (1 << 20)).to(torch.int32)


def _lex_unique(keys: torch.Tensor) -> torch.Tensor:
    """Sorted unique rows of an int32 [n,3] key tensor (deterministic, identical on every rank)."""
    if keys.shape[0] == 0:
        return keys
    return _unpack_keys(torch.unique(_pack_keys(keys), sorted=True))


def _as_tensor(keys):
    return keys if torch.is_tensor(keys) else torch.from_numpy(np.ascontiguousarray(keys))


def _canonical_keys_gather(volume, group=None, always_collective: bool = False, marks=None):
    """(Fallback of `canonical_keys` for blocks outside the bitmap window; the round-4 exchange.)
    Union of the block keys of all ranks in canonical order, on the local device, + the OR of the ranks'
    overflow flags + an upper bound of every voxel weight of the SUM over the ranks + whether any rank holds an already
    all-reduced (replicated) state.  One fixed-size all_gather (persistent buffers); one host read (of the gathered header
    rows).  Halo copies held by a volume are not its blocks (sentinel keys).
    Weight bound: a rank's state = a part it inherited (`frames_base`: a reduce-scatter leaves the ranks with DISJOINT parts
    of one reduced volume, so across ranks the inherited bounds do not add up -- their maximum holds) + the frames it
    integrated since (`frames_local`: these do add up).
    The exchange is latency, not bytes (C2: 3960 keys): the local list is written straight into the send buffer, the header
    travels as ONE small copy, the ranks' lists are cut out of the gathered buffer by the counts of the header (a mask would
    cost a device -> host round trip), sentinel rows are only filtered when some rank says it holds halo copies."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    collective = world > 1 or (always_collective and dist.is_initialized())
    if not collective:
        keys = _as_tensor(volume.block_keys(raise_on_overflow=False))
        _, _, ov = volume.status(raise_on_overflow=False)
        return _lex_unique(keys[keys[:, 0] != _SENTINEL]), int(ov), 0, int(volume.frames_integrated), False
    K = int(volume.max_blocks)
    n_local, _, ov = volume.status(raise_on_overflow=False)
    n_local = min(int(n_local), K)
    buf = volume.exchange_buffer("keys_send", (K + 2, 3), torch.int32, volume.exchange_device())
    volume.block_keys(out=buf, n=n_local)
    flags = int(bool(volume.replicated)) | (2 if volume.has_halo else 0)
    head_local = torch.tensor([[n_local, int(ov), K], [int(volume.frames_local + volume.frames_injected), int(volume.frames_base), flags]], dtype=torch.int32)
    buf[K:].copy_(head_local)
    gathered = volume.exchange_buffer("keys_recv", (world * (K + 2), 3), torch.int32, buf.device)
    _mark(marks, "keys: local list + header")
    _all_gather_into_tensor(gathered, buf, group)
    _mark(marks, "keys: all_gather")
    g = gathered.view(world, K + 2, 3)
    head = g[:, K:, :].cpu().numpy().astype(np.int64)         # the one host read of the exchange; [world, 2, 3]
    _mark(marks, "keys: header read")
    if int(head[:, 0, 2].min()) != K or int(head[:, 0, 2].max()) != K:
        raise RuntimeError("reduce_volume: every rank must create its volume with the same max_blocks")
    ov_any = int(np.bitwise_or.reduce(head[:, 0, 1]))
    frames_total = int(head[:, 1, 0].sum()) + int(head[:, 1, 1].max())
    flags_any = int(np.bitwise_or.reduce(head[:, 1, 2]))
    replicated = bool(flags_any & 1) and world > 1
    counts = [min(max(int(c), 0), K) for c in head[:, 0, 0]]
    parts = [g[r, :c] for r, c in enumerate(counts) if c]
    if not parts:
        return g[0, :0], ov_any, 1, frames_total, replicated
    body = parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)
    if flags_any & 2:
        body = body[body[:, 0] != _SENTINEL]
    _mark(marks, "keys: host checks + slices")
    return _lex_unique(body), ov_any, 1, frames_total, replicated



def canonical_keys(volume, group=None, always_collective: bool = False, marks=None, keys_via: str = "map"):
    """Union of the block keys of all ranks in canonical order (ascending (x, y, z)), on the local device, + the OR of the ranks'
    overflow flags + the number of collectives used + an upper bound of every voxel weight of the SUM over the ranks + whether
    any rank holds an already all-reduced (replicated) state.

    Block-map exchange: ``gs2m_tsdf_block_map`` marks this rank's blocks in a dense map over the
    volume's exchange window, one BYTE per block (RCCL has no bitwise reduction: 256 KiB for the default 64^3-block window), with
    the header bytes behind it; ONE ``all_reduce(MAX)`` over uint8 merges the ranks; ``gs2m_tsdf_map_keys`` turns the marked
    cells into the key list in cell order -- identical on every rank by construction: no gather of key lists, no sort / unique,
    no host pass over keys; one host read (the 32 + 8 R header bytes with the key count).  A block outside the window on ANY
    rank shows in the reduced header: every rank then takes the gather path (`_canonical_keys_gather`) together.
    Weight bound: a rank's state = a part it inherited (`frames_base`: a reduce-scatter leaves the ranks with DISJOINT parts of
    one reduced volume, so across ranks the inherited bounds do not add up -- their maximum holds) + the frames it integrated
    since (`frames_local`: these do add up)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    collective = world > 1 or (always_collective and dist.is_initialized())
    if not collective or keys_via == "gather":
        return _canonical_keys_gather(volume, group, always_collective, marks)
    K = int(volume.max_blocks)
    dev = volume.exchange_device()
    _agree_on_window(volume, group, world, dev)
    nb = volume.map_bytes(world)
    cells = volume.exchange_buffer("block_map", (nb,), torch.uint8, dev)
    flags = int(bool(volume.replicated)) | (2 if volume.has_halo else 0)
    volume.block_map(cells, rank, world, flags)
    _mark(marks, "keys: local block map")
    _all_reduce(cells, dist.ReduceOp.MAX, group)
    _mark(marks, "keys: all_reduce(MAX, u8)")
    n_cells = nb - _MAP_HEADER - 8 * world
    kbuf = volume.exchange_buffer("keys_union", (min(world * K, n_cells), 3), torch.int32, dev)
    head = volume.map_keys(cells, world, kbuf).astype(np.int64)
    _mark(marks, "keys: map -> keys + header read")

    def u32(off):
        return int(head[off]) | int(head[off + 1]) << 8 | int(head[off + 2]) << 16 | int(head[off + 3]) << 24

    if any(int(head[8 + i]) + int(head[12 + i]) != 255 for i in range(4)):
        raise RuntimeError("reduce_volume: every rank must create its volume with the same max_blocks")
    if any(int(head[16 + i]) + int(head[20 + i]) != 255 for i in range(4)):
        raise RuntimeError("reduce_volume: every rank must use the same exchange window (set_exchange_window)")
    if head[0]:
        # some rank holds a block outside the window: everybody saw the same byte, everybody gathers
        keys, ov_any, n_coll, frames_total, replicated = _canonical_keys_gather(volume, group, always_collective, marks)
        return keys, ov_any, n_coll + 1, frames_total, replicated
    frames_total = sum(u32(_MAP_HEADER + 8 * r) for r in range(world)) + max(u32(_MAP_HEADER + 8 * r + 4) for r in range(world))
    n = u32(24)
    if n > kbuf.shape[0]:
        raise RuntimeError(f"reduce_volume: the union holds {n} blocks, more than the ranks' pools together")
    ov_any = sum((1 << b) for b in range(4) if head[1 + b])
    # a copy: `kbuf` is the volume's persistent exchange buffer and the next reduction overwrites it in place, while the
    # result dict of this one (`info["keys"]`: exchange_halo, owner bookkeeping) may be kept (a few KB .. MB)
    return kbuf[:n].clone(), ov_any, 1, frames_total, bool(head[5]) and world > 1


def _agree_on_window(volume, group, world, dev):
    """The block-map all_reduce needs the SAME buffer size on every rank: windows of different `dim` (or volumes of different
    `max_blocks`) would enter the collective with mismatched tensors -- a hang or undefined behaviour under RCCL before the
    window hash in the reduced header could be checked.  One tiny fixed-size collective settles it, once per
    (window, max_blocks, group): MAX over (x, -x) of the seven integers = their maximum and minimum over the ranks."""
    key = (volume.exchange_window, int(volume.max_blocks), world, id(group))
    if getattr(volume, "_window_agreed", None) == key or world <= 1:
        return
    lo, dim = volume.exchange_window
    vals = [int(x) for x in (*lo, *dim, volume.max_blocks)]
    t = torch.tensor(vals + [-x for x in vals], dtype=torch.int64, device=dev)
    _all_reduce(t, dist.ReduceOp.MAX, group)
    got = t.cpu().tolist()
    if any(got[i] != -got[7 + i] for i in range(7)):
        what = "max_blocks" if all(got[i] == -got[7 + i] for i in range(6)) else "exchange window (set_exchange_window)"
        raise RuntimeError(f"reduce_volume: every rank must use the same {what}")
    volume._window_agreed = key


def reduce_volume(volume, group=None, mode: str = "reduce_scatter", always_collective: bool = False,
                  payload: str = "auto", algo: str = "rccl", marks=None, keys_via: str = "map"):
    """Sum-reduce the TSDF accumulators of all ranks into `volume`.

    mode "reduce_scatter": rank r ends with blocks [lo_r, hi_r) of the canonical list only (use `exchange_halo`
    before extracting its part of the mesh).  mode "allreduce": every rank ends with the complete fused volume.
    payload "packed" | "f32" | "auto" (packed when all ranks together integrated <= 1023 frames).
    algo "rccl" (the library's reduce-scatter / all-reduce) | "direct" (reduce_scatter only: one all_to_all of the
    1/R slices + a local sum over the R received copies).
    ``always_collective``: issue the collectives even at world size 1 (exercises the RCCL calls on one GPU).
    The caller must have drained the streams that integrate into `volume` (`RenderFusePipeline.drain`).
    Returns dict(n_blocks_union, bytes_per_rank, keys, owned, collectives, seconds, payload, algo, frames_total)."""
    t0 = time.perf_counter()
    from . import _lib
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if mode not in ("reduce_scatter", "allreduce") or payload not in ("auto", "packed", "f32") or algo not in ("rccl", "direct"):
        raise ValueError((mode, payload, algo))
    _mark(marks, "start")
    if keys_via not in ("map", "gather"):
        raise ValueError(keys_via)
    keys, ov_any, n_coll, frames_total, replicated = canonical_keys(volume, group, always_collective, marks, keys_via)
    _mark(marks, "keys: union")
    if replicated:
        # identical on every rank (it travelled in the gathered header): all of them raise
        raise RuntimeError("reduce_volume: a rank holds an all-reduced (replicated) volume -- summing it again would count its "
                           "frames once per rank; reset() it, or reduce with mode='reduce_scatter' between rounds")
    if ov_any:
        # every rank sees the same flags after the key exchange: all of them raise, nobody waits in a collective
        what = [n for b, n in _OVERFLOW_TEXT if ov_any & b]
        raise RuntimeError("TSDF volume overflow on at least one rank: " + ", ".join(what))
    fits = frames_total <= _lib.XFORM_PACKED_MAX_FRAMES
    if payload == "packed" and not fits:
        raise RuntimeError(f"packed exchange payload needs <= {_lib.XFORM_PACKED_MAX_FRAMES} frames in total, the ranks "
                           f"integrated {frames_total} (use payload='f32' or 'auto')")
    packed = payload == "packed" or (payload == "auto" and fits)
    form = _lib.XFORM_SUM_PACKED if packed else _lib.XFORM_SUM_F32
    collective = world > 1 or (always_collective and dist.is_initialized())
    n = int(keys.shape[0])
    dev = keys.device
    scatter = mode == "reduce_scatter" and collective
    n_pad = (n + world - 1) // world * world if scatter else n
    kpad = keys
    if n_pad != n:
        # pad with a key nobody owns (far outside any scene): packs to zeros, never unpacked
        filler = torch.full((n_pad - n, 3), _SENTINEL, dtype=torch.int32, device=dev)
        kpad = torch.cat([keys, filler], dim=0).contiguous()
    planes = 1 if packed else 5
    fbuf = volume.exchange_buffer("send_f32", (n_pad, planes, 4096), torch.float32, dev)
    ibuf = volume.exchange_buffer("send_i64", (n_pad, 4096), torch.int64, dev) if packed else None
    if n_pad:
        volume.pack(kpad, form, fbuf, ibuf)
    _mark(marks, "pack")
    nbytes = n_pad * 4096 * (12 if packed else 20)
    lo, hi = 0, n
    if collective and n_pad:
        if scatter:
            per = n_pad // world
            fout = volume.exchange_buffer("recv_f32", (per, planes, 4096), torch.float32, dev)
            iout = volume.exchange_buffer("recv_i64", (per, 4096), torch.int64, dev) if packed else None
            if algo == "direct":
                # slice p of every rank's buffer goes straight to rank p (one all_to_all per dtype), which sums its R copies
                fall = volume.exchange_buffer("a2a_f32", (world, per, planes, 4096), torch.float32, dev)
                _all_to_all_single(fall.view(-1), fbuf.view(-1), group)
                torch.sum(fall, dim=0, out=fout)
                n_coll += 1
                if packed:
                    iall = volume.exchange_buffer("a2a_i64", (world, per, 4096), torch.int64, dev)
                    _all_to_all_single(iall.view(-1), ibuf.view(-1), group)
                    torch.sum(iall, dim=0, out=iout)
                    n_coll += 1
            else:
                _reduce_scatter_tensor(fout, fbuf, group)
                n_coll += 1
                if packed:
                    _reduce_scatter_tensor(iout, ibuf, group)
                    n_coll += 1
            fbuf, ibuf = fout, iout
            lo, hi = min(n, rank * per), min(n, (rank + 1) * per)
            kpad = kpad[rank * per: (rank + 1) * per]
        else:
            _all_reduce(fbuf, dist.ReduceOp.SUM, group)
            n_coll += 1
            if packed:
                _all_reduce(ibuf, dist.ReduceOp.SUM, group)
                n_coll += 1
    _mark(marks, "collectives")
    if packed:
        # device-side check of the packed form (k_tsdf_pack): a local weight / colour sum that did not fit its field means the
        # frame bound was wrong (state injected through the C API).  The flag is LOCAL to the rank that packed the voxel, the
        # corrupted (field-carry) sums reach every rank: the verdict is made global with one 4-byte MAX all_reduce AFTER the
        # payload collectives, and every rank raises together -- none goes on to unpack, none waits in a later collective
        # (exchange_halo) for a rank that left.
        # (round 6: the flag word goes device -> device and is reduced there: ONE host read for the verdict instead of a status
        # synchronisation followed by the read of the reduced flag)
        flag = volume.exchange_buffer("pack_flags", (1,), torch.int32, dev)
        volume.flags_device(flag)
        if collective and world > 1:
            flag &= 8
            _all_reduce(flag, dist.ReduceOp.MAX, group)
            n_coll += 1
        bad = int(flag.item()) & 8
        if bad:
            raise RuntimeError("reduce_volume: " + _OVERFLOW_TEXT[3][1] + " (on at least one rank) -- the reduced buffers are "
                               "invalid; use payload='f32'")
    # replace the local state by the reduced blocks (reset only clears the slots in use)
    cnt = max(0, hi - lo)
    if cnt:
        # reset + unpack in one call: the slots the reduced blocks land in are not cleared first (C4: 1.2 GB of writes less)
        volume.replace(kpad[:cnt].contiguous(), form, fbuf[:cnt], ibuf[:cnt] if packed else None, frames=frames_total)
    else:
        volume.reset()
    volume.status()
    _mark(marks, "reset + unpack + status")
    volume.frames_base, volume.frames_local, volume.frames_injected = frames_total, 0, 0     # every weight of the reduced state is bounded by the total
    volume.replicated = collective and not scatter and world > 1
    return dict(n_blocks_union=n, bytes_per_rank=nbytes, keys=keys, owned=(lo, hi), collectives=n_coll, mode=mode,
                per=(n_pad // world if scatter else n), seconds=time.perf_counter() - t0,
                payload=("packed" if packed else "f32"), algo=(algo if scatter else "rccl"), frames_total=frames_total)


def exchange_window_from_cameras(camera_centers, max_depth: float, voxel_length: float, block: int = 16):
    """Window of block indices for `ScalableTSDFVolume.set_exchange_window` that covers everything the given cameras can put
    into the volume: the bounding box of the camera centres grown by the depth truncation (TSDF.run: baseline x
    TSDF_max_depth_baselines / TSDF_scale, tsdf_utils.py:86) plus one block.  Every rank knows every pose (views are sharded by
    index), so every rank computes the same window without talking to the others.  -> (lo[3], dim[3])"""
    c = np.asarray(camera_centers, np.float64).reshape(-1, 3)
    unit = float(voxel_length) * block
    lo = np.floor((c.min(axis=0) - max_depth) / unit).astype(np.int64) - 1
    hi = np.floor((c.max(axis=0) + max_depth) / unit).astype(np.int64) + 1
    return tuple(int(x) for x in lo), tuple(int(x) for x in (hi - lo + 1))


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
    from . import _lib
    # the blocks are already reduced: they travel VERBATIM (tsdf, weight, colour sums), so that a seam voxel has the same
    # bits on both sides (tsdf * w / w is not always tsdf)
    sbuf = volume.exchange_buffer("halo_send", (max(int(s_all.numel()), 1), 5, 4096), torch.float32, dev)
    if s_all.numel():
        volume.pack(keys[s_all].contiguous(), _lib.XFORM_RAW_F32, sbuf[: s_all.numel()])
    rbuf = volume.exchange_buffer("halo_recv", (max(int(r_all.numel()), 1), 5, 4096), torch.float32, dev)
    _all_to_all_single(rbuf[: r_all.numel()].reshape(-1), sbuf[: s_all.numel()].reshape(-1), group,
                       output_split_sizes=[c * 5 * 4096 for c in recv_counts],
                       input_split_sizes=[c * 5 * 4096 for c in send_counts])
    if r_all.numel():
        volume.unpack(keys[r_all].contiguous(), _lib.XFORM_RAW_F32, rbuf[: r_all.numel()], halo=True)
    volume.status()
    return int(r_all.numel())
