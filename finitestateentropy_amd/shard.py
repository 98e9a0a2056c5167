"""Block sharding across the GPUs of one node (one process per GPU, torch.distributed; "nccl" = RCCL on ROCm).

Every 32 KB block is coded with its own table and shares no state with its neighbours (the reference's
serial chunk loop, programs/bench.c:353-364,389-424, has no carried dependence), so the path shards by
contiguous block ranges with NO collective on the data path.  Collectives appear only where a caller keeps
the corpus on one rank: `scatter_blocks` (root -> ranks) before and `gather_blocks` (ranks -> root) after.
Both use fixed-stride slots like the reference bench's output buffers (programs/bench.c:514-516,545), one
point-to-point transfer per peer so that a root drives its xGMI links concurrently (a star, not a ring).

Everything here is backend-agnostic (`gloo` on CPU tensors in the tests, `nccl`/RCCL on device tensors).
"""
import torch
import torch.distributed as dist


def _staged(group=None):
    """point-to-point transfers of device tensors go through host memory when the process group cannot move them itself
    (gloo: the CPU tests, and two ranks sharing one GPU in a smoke test); RCCL moves device memory directly over xGMI"""
    return dist.get_backend(group) == "gloo"


def _grouped(ops, group=None):
    """Issue a list of (kind, tensor, peer) transfers as ONE group and wait for all of them: `dist.batch_isend_irecv` is
    ncclGroupStart ... ncclGroupEnd on RCCL, so a root's transfers to / from its peers are in flight together and its
    xGMI links run concurrently instead of one after the other.  Transfers between one pair of ranks match in list order on both
    sides.  Device tensors on a gloo group are staged through host memory (tests / two ranks sharing a GPU)."""
    if not ops:
        return
    stage = _staged(group)
    p2p, landing = [], []
    for kind, t, peer in ops:
        if kind == "send":
            p2p.append(dist.P2POp(dist.isend, t.cpu() if (t.is_cuda and stage) else t, peer, group))
        elif t.is_cuda and stage:
            h = torch.empty(t.shape, dtype=t.dtype)
            landing.append((t, h))
            p2p.append(dist.P2POp(dist.irecv, h, peer, group))
        else:
            p2p.append(dist.P2POp(dist.irecv, t, peer, group))
    for q in dist.batch_isend_irecv(p2p):
        q.wait()
    for t, h in landing:
        t.copy_(h)


def shard_range(n_blocks, rank, world):
    """Contiguous range [lo, hi) of rank `rank`; sizes differ by at most one block."""
    base, rem = divmod(n_blocks, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def scatter_blocks(blocks_root, n_blocks, block_bytes, rank, world, device, root=0, group=None, out=None):
    """Root holds (n_blocks, block_bytes) uint8; every rank receives its shard_range rows (into `out` if given).  The root's
    sends to all peers form one group (one transfer per peer, rows of a contiguous range are contiguous memory)."""
    lo, hi = shard_range(n_blocks, rank, world)
    mine = out if out is not None else torch.empty((hi - lo, block_bytes), dtype=torch.uint8, device=device)
    assert mine.shape == (hi - lo, block_bytes) and mine.is_contiguous()
    if world == 1:
        mine.copy_(blocks_root[lo:hi])
        return mine
    if rank == root:
        ops = []
        for r in range(world):
            rlo, rhi = shard_range(n_blocks, r, world)
            if r == root:
                mine.copy_(blocks_root[rlo:rhi])
            elif rhi > rlo:
                ops.append(("send", blocks_root[rlo:rhi].contiguous(), r))
        _grouped(ops, group)
    elif hi > lo:
        _grouped([("recv", mine, root)], group)
    return mine


def gather_blocks(slots_mine, sizes_mine, n_blocks, rank, world, root=0, group=None, out=None):
    """Fixed-stride result slots (rows, stride) + per-block result values -> root gets (n_blocks, stride) and sizes (in `out` =
    (slots, sizes) if given).  All transfers of the root -- two per peer, slots then sizes -- form one group.
    Non-root ranks return (None, None)."""
    stride = slots_mine.shape[1]
    if world == 1:
        return slots_mine, sizes_mine
    if rank == root:
        if out is not None:
            slots, sizes = out
        else:
            slots = torch.empty((n_blocks, stride), dtype=slots_mine.dtype, device=slots_mine.device)
            sizes = torch.empty(n_blocks, dtype=sizes_mine.dtype, device=sizes_mine.device)
        ops = []
        for r in range(world):
            rlo, rhi = shard_range(n_blocks, r, world)
            if r == root:
                slots[rlo:rhi].copy_(slots_mine); sizes[rlo:rhi].copy_(sizes_mine)
            elif rhi > rlo:
                ops.append(("recv", slots[rlo:rhi], r))
                ops.append(("recv", sizes[rlo:rhi], r))
        _grouped(ops, group)
        return slots, sizes
    if slots_mine.shape[0]:
        _grouped([("send", slots_mine.contiguous(), root), ("send", sizes_mine.contiguous(), root)], group)
    return None, None


def max_over_ranks(values, device, world, group=None):
    """bench.py timing rule: the slowest rank defines the step time."""
    t = torch.tensor(values, dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return [float(x) for x in t.tolist()]


def sharded_roundtrip(blocks_root, n_blocks, block_bytes, rank, world, device, compress_fn, decompress_fn, root=0):
    """scatter -> per-rank compress -> per-rank decompress -> gather (compressed slots + sizes, decoded blocks).
    compress_fn(blocks) -> (slots, sizes); decompress_fn(slots, sizes, block_bytes) -> (blocks, results)."""
    mine = scatter_blocks(blocks_root, n_blocks, block_bytes, rank, world, device, root)
    slots, sizes = compress_fn(mine)
    back, results = decompress_fn(slots, sizes, block_bytes)
    g_slots, g_sizes = gather_blocks(slots, sizes, n_blocks, rank, world, root)
    g_back, g_res = gather_blocks(back, results, n_blocks, rank, world, root)
    return g_slots, g_sizes, g_back, g_res


def sharded_codec_job(corpus_root, n_blocks, block_bytes, rank, world, device, codecs, root=0, shard_out=None, gather_out=None, mark=None):
    """BASELINE config 5 with the corpus on one rank (bench.py's with-comm variant): scatter the raw blocks, run every codec's
    encode + decode on the rank's shard, gather every codec's compressed slots and sizes on the root.
    `codecs`: objects with .src (set here), .encode(), .decode(), .dst (rows, stride), .res, .out, .dres.
    `shard_out` / `gather_out` (one (slots, sizes) pair per codec, root only): preallocated landing buffers, so that a timed pass
    does not allocate; `mark(name)` is called after each phase (bench.py's per-phase clock).  Returns (my shard, [(slots, sizes) per codec on the root, (None, None) elsewhere])."""
    mine = scatter_blocks(corpus_root, n_blocks, block_bytes, rank, world, device, root, out=shard_out)
    if mark:
        mark("1_scatter")
    for cd in codecs:
        cd.src = mine
        cd.encode()
        cd.decode()
    if mark:
        mark("2_codecs")
    gathered = [gather_blocks(cd.dst, cd.res, n_blocks, rank, world, root, out=(gather_out[i] if gather_out is not None else None))
                for i, cd in enumerate(codecs)]
    if mark:
        mark("3_gather")
    return mine, gathered


def sharded_job_ok(mine, gathered, codecs, n_blocks, block_bytes, rank, world, root=0):
    """round trip on this rank's shard, and (root) the gathered sizes of its own range equal its local results"""
    ok = all(bool((cd.dres == block_bytes).all()) and torch.equal(cd.out, mine) for cd in codecs)
    if rank == root:
        lo, hi = shard_range(n_blocks, rank, world)
        ok = ok and all(torch.equal(g[1][lo:hi], cd.res) for g, cd in zip(gathered, codecs))
    return ok
