"""N>1 path on CPU: world_size-2 `gloo` processes exercise the shard / scatter / gather plumbing of
finitestateentropy_amd.shard with the CPU oracle standing in for the device codec (test infrastructure only)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_blocks, block_bytes, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from finitestateentropy_amd import shard
    from oracle.oracle import Oracle
    orc = Oracle()
    blocks_root = None
    if rank == 0:
        mix = [orc.probagen_batch(P, n_blocks // 3 + 1, block_bytes, 1)[:n_blocks // 3 + 1] for P in (2, 14, 80)]
        allb = np.empty((n_blocks, block_bytes), np.uint8)
        for b in range(n_blocks):
            allb[b] = mix[b % 3][b // 3]
        blocks_root = torch.from_numpy(allb)

    def comp(x):
        _, res, dst = orc.compress_batch(0, x.numpy(), table_log=11, nthreads=1)
        return torch.from_numpy(dst), torch.from_numpy(res.astype(np.int64))

    def decomp(slots, sizes, bs):
        _, res, out = orc.decompress_batch(0, slots.numpy(), sizes.numpy().astype(np.uint64), bs, nthreads=1)
        return torch.from_numpy(out), torch.from_numpy(res.astype(np.int64))

    # bench.py's N>1 code path (config 5 with the corpus on rank 0): the very function bench.py calls, FSE and Huff0 codecs
    class CpuCodec:
        def __init__(self, codec):
            self.codec = codec; self.src = None
        def encode(self):
            _, res, dst = orc.compress_batch(self.codec, self.src.numpy(), table_log=11, nthreads=1)
            self.dst, self.res = torch.from_numpy(dst), torch.from_numpy(res.astype(np.int64))
        def decode(self):
            _, res, out = orc.decompress_batch(self.codec, self.dst.numpy(), self.res.numpy().astype(np.uint64), block_bytes, nthreads=1)
            self.out, self.dres = torch.from_numpy(out), torch.from_numpy(res.astype(np.int64))
    cds = [CpuCodec(0), CpuCodec(1)]
    mine, gathered = shard.sharded_codec_job(blocks_root, n_blocks, block_bytes, rank, world, "cpu", cds)
    job_ok = shard.sharded_job_ok(mine, gathered, cds, n_blocks, block_bytes, rank, world)
    lo, hi = shard.shard_range(n_blocks, rank, world)
    assert mine.shape[0] == hi - lo
    if rank == 0:
        for codec, (slots, sizes) in zip((0, 1), gathered):
            _, sres, sdst = orc.compress_batch(codec, blocks_root.numpy(), table_log=11, nthreads=1)
            job_ok = job_ok and bool((sizes.numpy() == sres.astype(np.int64)).all()) and all(
                bool((slots[b, :int(sres[b])].numpy() == sdst[b][:int(sres[b])]).all()) for b in range(n_blocks))
    else:
        job_ok = job_ok and all(g == (None, None) for g in gathered)
    assert job_ok

    # the pipelined variant with variable-length (packed) results: pieces of every shard, sizes exchanged before each gather
    def cpu_compact(pc, src):
        recs = []
        for b in range(src.shape[0]):
            r = int(pc.res[b])
            recs.append(src[b].numpy() if r == 0 else src[b].numpy()[:1] if r == 1 else pc.dst[b].numpy()[:r])
        lens = np.array([len(x) for x in recs], np.int64)
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        return torch.from_numpy(np.concatenate(recs) if len(recs) else np.zeros(0, np.uint8)), torch.from_numpy(offs)

    class CpuPieces(CpuCodec):
        def piece(self, lo, hi):
            return CpuCodec(self.codec)
    for pieces in (1, 4, 7):
        cdp = [CpuPieces(0), CpuPieces(1)]
        mine2, packed_g, stats = shard.sharded_codec_job_pipelined(blocks_root, n_blocks, block_bytes, rank, world, "cpu", cdp, cpu_compact, pieces=pieces)
        assert torch.equal(mine2, mine)
        if rank == 0:
            payload = 0
            for codec, (pk, of) in zip((0, 1), packed_g):
                _, sres, sdst = orc.compress_batch(codec, blocks_root.numpy(), table_log=11, nthreads=1)
                of = of.numpy(); pk = pk.numpy()
                assert of[0] == 0 and len(stats["order"]) == n_blocks and sorted(stats["order"]) == list(range(n_blocks))
                for row, b in enumerate(stats["order"]):                 # the records of the packed order, byte for byte
                    r = int(sres[b])
                    rec = blocks_root[b].numpy() if r == 0 else blocks_root[b].numpy()[:1] if r == 1 else sdst[b][:r]
                    assert of[row + 1] - of[row] == len(rec) and (pk[of[row]:of[row + 1]] == rec).all(), (pieces, codec, b)
                payload += int(of[n_blocks])
            lo0, hi0 = shard.shard_range(n_blocks, 0, world)
            own = sum(int(of_[n_blocks]) for _, of_ in packed_g)
            assert stats["payload_bytes"] == payload
            assert stats["scatter_bytes"] == (n_blocks - (hi0 - lo0)) * block_bytes
            # what crossed the links on the way back: the peers' records plus 8 bytes per offset entry (rows + 1 per peer, piece and codec)
            assert stats["gather_bytes"] <= payload + 8 * 2 * (n_blocks + pieces * world) and stats["gather_bytes"] >= 8 * 2 * (n_blocks - (hi0 - lo0))
        else:
            assert all(g == (None, None) for g in packed_g)

    g_slots, g_sizes, g_back, g_res = shard.sharded_roundtrip(blocks_root, n_blocks, block_bytes, rank, world, "cpu", comp, decomp)
    t = shard.max_over_ranks([float(rank + 1), 0.5], "cpu", world)
    assert t == [float(world), 0.5]
    if rank == 0:
        s_slots, s_sizes = comp(blocks_root)
        ok = bool((g_sizes == s_sizes).all()) and all(
            bool((g_slots[b, :int(s_sizes[b])] == s_slots[b, :int(s_sizes[b])]).all()) for b in range(n_blocks))
        ok = ok and bool((g_res == block_bytes).all()) and bool((g_back == blocks_root).all())
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges():
    sys.path.insert(0, ROOT)
    from finitestateentropy_amd.shard import shard_range
    for n in (0, 1, 7, 8, 100000, 1000003):
        for w in (1, 2, 3, 4, 8):
            edges = [shard_range(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


def _run_world(world, n_blocks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_blocks, 4096, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_scatter_compress_gather_world2():
    _run_world(2, 37)


def test_scatter_compress_gather_world3_uneven_shards():
    """three ranks: the root's grouped scatter / gather has two peers, the shards are 14 + 13 + 13 blocks"""
    _run_world(3, 40)
