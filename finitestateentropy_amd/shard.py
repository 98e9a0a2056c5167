"""Block sharding across the GPUs of one node (one process per GPU, torch.distributed; "nccl" = RCCL on ROCm).

Every 32 KB block is coded with its own table and shares no state with its neighbours (the reference's
serial chunk loop, programs/bench.c:353-364,389-424, has no carried dependence), so the path shards by
contiguous block ranges with NO collective on the data path.  Collectives appear only where a caller keeps
the corpus on one rank: `scatter_blocks` (root -> ranks) before and `gather_blocks` (ranks -> root) after.
Both use fixed-stride slots like the reference bench's output buffers (programs/bench.c:514-516,545), one
point-to-point transfer per peer so that a root drives its xGMI links concurrently (a star, not a ring).

Everything here is backend-agnostic (`gloo` on CPU tensors in the tests, `nccl`/RCCL on device tensors).

Streams (the pipelined job, RCCL semantics): a transfer posted with `dist.batch_isend_irecv` starts after the work already queued on the
stream that is CURRENT when it is posted, and `Work.wait()` makes the CURRENT stream wait for it (the host does not block).  The
pipelined job therefore posts and waits for its transfers on two side streams ("lanes": one per direction) that depend on the compute
stream only through events -- the compute stream itself waits for exactly one thing per piece, the arrival of that piece's raw blocks --
and exchanges the packed sizes as HOST integers over a gloo side group, so that nothing between "codecs of piece k queued" and "gather
of piece k - 1 posted" drains the compute stream.  The few stream / event / process-group calls sit behind the `_`-prefixed functions
below so that tests/test_shard_streams.py can swap them for recorders and check the order of what is issued without a GPU.
"""
import contextlib

import torch
import torch.distributed as dist


# ---- stream plumbing: no-ops for CPU tensors
def _is_cuda(device):
    return torch.device(device).type == "cuda"


def _new_stream(device, name):
    """a side stream (`name` is for the recorders of the tests)"""
    return torch.cuda.Stream(device) if _is_cuda(device) else None


def _on(stream):
    return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()


def _record_event(device, what):
    """an event behind everything queued so far on the current stream"""
    if not _is_cuda(device):
        return None
    ev = torch.cuda.Event()
    ev.record()
    return ev


def _stream_wait_event(stream, ev):
    if stream is not None and ev is not None:
        stream.wait_event(ev)


def _current_wait_event(ev):
    if ev is not None:
        torch.cuda.current_stream().wait_event(ev)


def _current_wait_stream(stream):
    if stream is not None:
        torch.cuda.current_stream().wait_stream(stream)


def _host_wait_event(ev):
    if ev is not None:
        ev.synchronize()


def _record_stream(t, stream):
    """a tensor allocated on the compute stream is about to be used on `stream`: tell the caching allocator"""
    if stream is not None and t is not None and t.is_cuda:
        t.record_stream(stream)


def _batch_p2p(ops, group):
    """(kind, tensor, peer) triples -> one ncclGroupStart ... ncclGroupEnd of sends / receives; returns the requests"""
    p2p = [dist.P2POp(dist.isend if kind == "send" else dist.irecv, t, peer, group) for kind, t, peer in ops]
    return dist.batch_isend_irecv(p2p) if p2p else []


_HOST_GROUPS = {}          # id(group) -> (group, its gloo side group): the entry is only trusted while it still holds the SAME group object


def host_group(group=None):
    """the process group the few HOST integers of the pipelined job travel over: `group` itself when it is gloo, else a gloo group over
    the same ranks, created once (collectively: every rank of `group` gets here at the same point of the job) and kept until
    `destroy_host_groups()` -- call that before destroying `group` (or the default group) if the process goes on to create others:
    the cache is keyed by the group object and checked for identity, so a new group that happens to get a dead one's id() never
    inherits its side group, but the side groups themselves are only released there."""
    if dist.get_backend(group) == "gloo":
        return group
    g = group if group is not None else dist.group.WORLD
    entry = _HOST_GROUPS.get(id(g))
    if entry is None or entry[0] is not g:
        ranks = None if group is None else dist.get_process_group_ranks(group)
        entry = (g, dist.new_group(ranks=ranks, backend="gloo"))
        _HOST_GROUPS[id(g)] = entry
    return entry[1]


def destroy_host_groups():
    """release every gloo side group host_group() created (collective over each of them, like their creation)"""
    for _, hg in list(_HOST_GROUPS.values()):
        try:
            dist.destroy_process_group(hg)
        except Exception:       # the default group is gone already: its side groups went with it
            pass
    _HOST_GROUPS.clear()


def host_totals(value, world, hgroup):
    """every rank's integer `value` on every rank, as python ints, over a host-side (gloo) group: the sizes that have to be known before
    variable-length transfers can be posted ("size exchange") -- without touching a device stream"""
    if world == 1:
        return [int(value)]
    t = torch.tensor([int(value)], dtype=torch.int64)
    out = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(out, t, group=hgroup)
    return [int(x[0]) for x in out]


def _staged(group=None):
    """point-to-point transfers of device tensors go through host memory when the process group cannot move them itself
    (gloo: the CPU tests, and two ranks sharing one GPU in a smoke test); RCCL moves device memory directly over xGMI"""
    return dist.get_backend(group) == "gloo"


def _stage_ops(ops, group):
    """device tensors on a gloo group are staged through host memory (tests / two ranks sharing a GPU): returns (ops, landing copies)"""
    if not _staged(group):
        return list(ops), []
    out, landing = [], []
    for kind, t, peer in ops:
        if kind == "send":
            out.append((kind, t.cpu() if t.is_cuda else t, peer))
        elif t.is_cuda:
            h = torch.empty(t.shape, dtype=t.dtype)
            landing.append((t, h))
            out.append((kind, h, peer))
        else:
            out.append((kind, t, peer))
    return out, landing


def _grouped(ops, group=None):
    """Issue a list of (kind, tensor, peer) transfers as ONE group and wait for all of them: `dist.batch_isend_irecv` is
    ncclGroupStart ... ncclGroupEnd on RCCL, so a root's transfers to / from its peers are in flight together and its
    xGMI links run concurrently instead of one after the other.  Transfers between one pair of ranks match in list order on both
    sides.  Device tensors on a gloo group are staged through host memory (tests / two ranks sharing a GPU)."""
    _finish(_post(ops, group))


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


def _post(ops, group=None):
    """like _grouped, but returns at once: (requests, landing copies to make after they complete).  With RCCL the transfers start behind
    what is queued on the CURRENT stream and a request's wait() makes the CURRENT stream wait for the transfer (no host block)."""
    if not ops:
        return [], []
    staged, landing = _stage_ops(ops, group)
    return _batch_p2p(staged, group), landing


def _finish(posted):
    reqs, landing = posted
    for q in reqs:
        q.wait()
    for t, h in landing:
        t.copy_(h)


def _total_later(offsets, device):
    """the packed size offsets[-1] as a host integer, without draining the stream: a non-blocking copy into pinned memory and an event
    recorded behind it now, the wait when the value is asked for (by then the device is busy with the next piece)"""
    if offsets.is_cuda:
        host = torch.empty(1, dtype=torch.int64).pin_memory()
        host.copy_(offsets[-1:], non_blocking=True)
    else:
        host = offsets[-1:]
    ev = _record_event(device, "total")

    def ready():
        _host_wait_event(ev)
        return int(host[0])
    return ready


def piece_ranges(lo, hi, pieces):
    """[lo, hi) cut into `pieces` contiguous ranges of nearly equal length (empty ones when there are fewer blocks than pieces)"""
    n = hi - lo
    return [(lo + (n * k) // pieces, lo + (n * (k + 1)) // pieces) for k in range(pieces)]


def post_gather_packed(packed_mine, offsets_mine, total_mine, totals, rows, rank, world, packed_root, offsets_root, base, root=0, group=None):
    """One variable-length gather, posted without waiting: every rank's packed records (`total_mine` bytes) and record offsets
    (rows[r] + 1 int64 each) land on the root -- bytes at packed_root[base + sum(totals[:r]) ...], offsets in scratch tensors.
    Returns (posted, fix) -- `fix()` after _finish(posted) rebases the offsets into offsets_root[row0 : row0 + rows[r]] per rank and
    returns the bytes this call moved over the links (payload + 8 per offset entry)."""
    wire = 0
    if rank == root:
        ops, scratch, pos = [], [], base
        for r in range(world):
            if r == root:
                scratch.append(None)
            else:
                tmp = torch.empty(rows[r][1] - rows[r][0] + 1, dtype=torch.int64, device=offsets_root.device)
                scratch.append(tmp)
                if totals[r]:
                    ops.append(("recv", packed_root[pos:pos + totals[r]], r))
                ops.append(("recv", tmp, r))
                wire += totals[r] + 8 * tmp.numel()
            pos += totals[r]
        posted = _post(ops, group)

        def fix():
            pos = base
            for r in range(world):
                r0, r1 = rows[r]
                if r == root:
                    packed_root[pos:pos + totals[r]].copy_(packed_mine[:totals[r]])
                    offsets_root[r0:r1].copy_(offsets_mine[:r1 - r0] + pos)
                else:
                    offsets_root[r0:r1].copy_(scratch[r][:r1 - r0] + pos)
                pos += totals[r]
            return wire
        return posted, fix
    ops = []
    if total_mine:
        ops.append(("send", packed_mine[:total_mine], root))
    ops.append(("send", offsets_mine, root))
    return _post(ops, group), (lambda: 0)


def sharded_codec_job_pipelined(corpus_root, n_blocks, block_bytes, rank, world, device, codecs, compact_fn, pieces=4, root=0, group=None,
                                shard_out=None, packed_out=None, offsets_out=None, gather_group=None):
    """BASELINE config 5 with the corpus on one rank, PIPELINED and with variable-length results: every rank's shard is cut into `pieces`
    contiguous pieces; the root's scatter of piece k + 1, every rank's codecs on piece k and the gather of piece k - 1 overlap.  What
    travels back is not fixed-stride slots but the packed records of FSEHIP_compact_batch plus their offsets: `compact_fn(codec_piece,
    src_piece) -> (packed uint8, offsets int64 (rows + 1))`; before a gather the ranks exchange their packed sizes (one integer each).
    `codecs`: objects with .piece(lo, hi) -> an object for rows [lo, hi) of this rank's shard with .src (assignable), .encode(), .decode(),
    .dst, .res, .out, .dres.  Returns (my shard, [(packed, offsets) per codec on the root | (None, None)], stats) where offsets has
    n_blocks + 1 entries (global block order inside every piece round: pieces are gathered rank after rank) and stats counts the bytes
    the root moved: scatter_bytes, gather_bytes, payload_bytes.

    Who waits for what (module docstring): three streams per rank -- the caller's COMPUTE stream (codecs, compaction), a SCATTER lane
    and a GATHER lane.  Transfers are posted and waited for on their lane; the lanes take their dependencies from the compute stream
    as events (scatter lane: the start of the job; gather lane: "piece k compacted"), the compute stream from the scatter lane ("piece
    k has landed") and, once at the end, joins both lanes.  The packed sizes travel as host integers over a gloo side group; the host
    waits only for the event behind piece k - 1's compaction, after piece k's kernels are queued.  `gather_group`: a second process
    group over the same ranks for the way back -- RCCL serialises the operations of one communicator on one stream, so with a
    communicator per direction the scatter of piece k + 1 and the gather of piece k - 1 really run side by side (default: `group`).
    Buffer lifetime: the scatter and gather lanes use `corpus_root`, the shard and the gathered buffers on their own streams WITHOUT
    record_stream -- the caller owns them and must keep them alive (and unmodified) until this function has returned: it joins both lanes
    into the compute stream before it does, so anything queued on the compute stream afterwards is ordered behind the transfers."""
    lo, hi = shard_range(n_blocks, rank, world)
    mine = shard_out if shard_out is not None else torch.empty((hi - lo, block_bytes), dtype=torch.uint8, device=device)
    ranges = [piece_ranges(*shard_range(n_blocks, r, world), pieces) for r in range(world)]      # [rank][piece] -> global rows
    stats = {"scatter_bytes": 0, "gather_bytes": 0, "payload_bytes": 0}
    ggroup = gather_group if gather_group is not None else group
    hgroup = host_group(group) if world > 1 else None
    lane_in, lane_out = _new_stream(device, "scatter"), _new_stream(device, "gather")
    start = _record_event(device, "start")                            # whatever used these buffers before (an earlier pass) is behind this
    _stream_wait_event(lane_in, start)
    _stream_wait_event(lane_out, start)

    def post_scatter(k):
        """piece k of every shard leaves the root, on the scatter lane; returns the event behind its arrival here"""
        with _on(lane_in):
            ops = []
            if world == 1:
                g0, g1 = ranges[0][k]
                mine[g0 - lo:g1 - lo].copy_(corpus_root[g0:g1])
            elif rank == root:
                for r in range(world):
                    g0, g1 = ranges[r][k]
                    if r == root:
                        mine[g0 - lo:g1 - lo].copy_(corpus_root[g0:g1])
                    elif g1 > g0:
                        ops.append(("send", corpus_root[g0:g1], r))
                        stats["scatter_bytes"] += (g1 - g0) * block_bytes
            else:
                g0, g1 = ranges[rank][k]
                if g1 > g0:
                    ops.append(("recv", mine[g0 - lo:g1 - lo], root))
            _finish(_post(ops, group))                                # (RCCL: the LANE waits, not the host)
            return _record_event(device, "landed %d" % k)

    gathered = None
    if rank == root:
        gathered = []
        for i in range(len(codecs)):
            pk = packed_out[i] if packed_out is not None else torch.empty(n_blocks * block_bytes, dtype=torch.uint8, device=device)
            of = offsets_out[i] if offsets_out is not None else torch.empty(n_blocks + 1, dtype=torch.int64, device=device)
            gathered.append((pk, of))
    bases = [0] * len(codecs)                                       # bytes of every codec's packed stream gathered so far
    row_base = 0                                                    # rows of the packed order gathered so far
    order = []                                                      # global block index of every row of the packed order (root)
    keep = []                                                       # per-piece tensors stay referenced until the lanes are joined

    def gather_piece(k, packs, compacted):
        """exchange the sizes of piece k (host integers), then post its variable-length transfers, wait for them and rebase the offsets --
        all on the gather lane, behind the event `compacted`"""
        nonlocal row_base
        rows, r0 = [], row_base
        for r in range(world):
            g0, g1 = ranges[r][k]
            rows.append((r0, r0 + (g1 - g0))); r0 += g1 - g0
        sized = []
        for packed, offsets, tot_ready in packs:
            total = tot_ready()                                      # host: waits for the event behind piece k's compaction only
            sized.append((total, host_totals(total, world, hgroup)))
        with _on(lane_out):
            _stream_wait_event(lane_out, compacted)
            for i, ((packed, offsets, _), (total, totals)) in enumerate(zip(packs, sized)):
                _record_stream(packed, lane_out); _record_stream(offsets, lane_out)
                pk, of = gathered[i] if rank == root else (None, None)
                posted, fix = post_gather_packed(packed, offsets, total, totals, rows, rank, world, pk, of, bases[i], root, ggroup)
                _finish(posted)                                      # (RCCL: the LANE waits, not the host and not the compute stream)
                stats["gather_bytes"] += fix()
                tot = sum(totals)
                bases[i] += tot
                stats["payload_bytes"] += tot
        if rank == root:
            for r in range(world):
                order.extend(range(*ranges[r][k]))
        row_base = r0

    landed = post_scatter(0)
    pending = None                                                  # (piece, packs, event) whose gather has not been posted yet
    for k in range(pieces):
        nxt = post_scatter(k + 1) if k + 1 < pieces else None
        _current_wait_event(landed)                                 # the one thing the compute stream waits for: its input
        g0, g1 = ranges[rank][k]
        packs = []
        for cd in codecs:
            pc = cd.piece(g0 - lo, g1 - lo)
            pc.src = mine[g0 - lo:g1 - lo]
            pc.encode(); pc.decode()
            packed, offsets = compact_fn(pc, pc.src)
            packs.append((packed, offsets, _total_later(offsets, device)))
        compacted = _record_event(device, "compacted %d" % k)
        keep.append(packs)
        if pending is not None:
            gather_piece(*pending)                                  # (its sizes are long known: the device is busy with piece k meanwhile)
        pending = (k, packs, compacted)
        landed = nxt
    gather_piece(*pending)
    _current_wait_stream(lane_in)
    _current_wait_stream(lane_out)                                  # the compute stream joins the lanes once, here
    if rank == root:
        for i in range(len(codecs)):
            gathered[i][1][n_blocks] = bases[i]
        stats["order"] = order
        return mine, gathered, stats
    return mine, [(None, None)] * len(codecs), stats


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
