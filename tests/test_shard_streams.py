"""Stream discipline of the pipelined with-comm job (finitestateentropy_amd.shard.sharded_codec_job_pipelined) checked WITHOUT a GPU and
without a process group: the module's stream / event / point-to-point plumbing is replaced by recorders, the job runs as one rank of a
pretended world, and the recorded order is asserted.

What RCCL semantics require for the gather of piece k - 1 to overlap the codecs of piece k (torch's ProcessGroupNCCL:
a transfer starts behind the work queued on the stream that is current when it is posted, Work.wait() makes the CURRENT stream wait):
  * no transfer is posted or waited for on the compute stream -- scatters live on the scatter lane, gathers on the gather lane;
  * the compute stream waits for exactly one event per piece ("piece k has landed") and joins the two lanes once, at the end;
  * the gather lane takes its dependency as an event recorded behind piece k - 1's compaction -- not behind piece k's kernels;
  * between "codecs of piece k queued" and "gather of piece k - 1 posted" the host waits only for events recorded BEFORE piece k's
    codecs were queued (the packed size of piece k - 1), and the sizes are exchanged as host integers (no device collective)."""
import contextlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class Recorder:
    def __init__(self):
        self.log = []
        self.current = "compute"

    def add(self, *entry):
        self.log.append(entry + (self.current,))
        return len(self.log) - 1


class FakeStream:
    def __init__(self, name):
        self.name = name


class FakeEvent:
    def __init__(self, what, stream, seq):
        self.what, self.stream, self.seq = what, stream, seq


class FakeWork:
    def __init__(self, rec, tag):
        self.rec, self.tag = rec, tag

    def wait(self):
        self.rec.add("wait", self.tag)


def _install(monkeypatch, shard, rec, world):
    @contextlib.contextmanager
    def on(stream):
        prev = rec.current
        rec.current = stream.name if stream is not None else prev
        try:
            yield
        finally:
            rec.current = prev

    def record_event(device, what):
        seq = rec.add("record", what)
        return FakeEvent(what, rec.current, seq)

    def batch_p2p(ops, group):
        kinds = sorted(set(k for k, _, _ in ops))
        for kind, t, peer in ops:
            if kind == "recv":
                t.zero_()                       # nothing arrives: the pretended peers send zeros
        rec.add("post", tuple(kinds), group)
        return [FakeWork(rec, group)]

    monkeypatch.setattr(shard, "_new_stream", lambda device, name: FakeStream(name))
    monkeypatch.setattr(shard, "_on", on)
    monkeypatch.setattr(shard, "_record_event", record_event)
    monkeypatch.setattr(shard, "_stream_wait_event", lambda stream, ev: rec.add("stream_wait_event", stream.name, ev.what))
    monkeypatch.setattr(shard, "_current_wait_event", lambda ev: rec.add("current_wait_event", ev.what))
    monkeypatch.setattr(shard, "_current_wait_stream", lambda stream: rec.add("current_wait_stream", stream.name))
    monkeypatch.setattr(shard, "_host_wait_event", lambda ev: rec.add("host_wait_event", ev.what, ev.seq))
    monkeypatch.setattr(shard, "_record_stream", lambda t, stream: None)
    monkeypatch.setattr(shard, "_batch_p2p", batch_p2p)
    monkeypatch.setattr(shard, "_staged", lambda group=None: False)
    monkeypatch.setattr(shard, "host_group", lambda group=None: "host-gloo")

    def host_totals(value, w, hgroup):
        assert hgroup == "host-gloo"
        rec.add("host_totals", int(value))
        return [int(value)] * w                  # the pretended peers report the same size
    monkeypatch.setattr(shard, "host_totals", host_totals)


class Piece:
    """a stand-in codec: "compresses" every block to its first 5 bytes"""
    def __init__(self, rec, name, k):
        self.rec, self.name, self.k, self.src = rec, name, k, None

    def encode(self):
        self.rec.add("encode", self.name, self.k)
        self.dst = self.src[:, :5].clone()
        self.res = torch.full((self.src.shape[0],), 5, dtype=torch.int64)

    def decode(self):
        self.rec.add("decode", self.name, self.k)
        self.out, self.dres = self.src.clone(), torch.full((self.src.shape[0],), self.src.shape[1], dtype=torch.int64)


class Codec:
    def __init__(self, rec, name):
        self.rec, self.name, self.k = rec, name, 0

    def piece(self, lo, hi):
        self.k += 1
        return Piece(self.rec, self.name, self.k - 1)


def _compact(rec):
    def fn(pc, src):
        rec.add("compact", pc.name, pc.k)
        n = src.shape[0]
        return pc.dst.reshape(-1).clone(), torch.arange(n + 1, dtype=torch.int64) * 5
    return fn


@pytest.mark.parametrize("rank", [0, 1])
@pytest.mark.parametrize("pieces", [1, 4])
def test_pipelined_job_keeps_communication_off_the_compute_stream(monkeypatch, rank, pieces):
    from finitestateentropy_amd import shard
    rec = Recorder()
    world, n_blocks, bb = 3, 50, 64
    _install(monkeypatch, shard, rec, world)
    corpus = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (n_blocks, bb), dtype=np.uint8)) if rank == 0 else None
    codecs = [Codec(rec, "fse"), Codec(rec, "huf")]
    mine, gathered, stats = shard.sharded_codec_job_pipelined(corpus, n_blocks, bb, rank, world, "cpu", codecs, _compact(rec), pieces=pieces,
                                                              group="grp-in", gather_group="grp-out")
    log = rec.log
    lo, hi = shard.shard_range(n_blocks, rank, world)
    if rank == 0:
        assert torch.equal(mine, corpus[lo:hi])

    # 1. every transfer is posted and waited for on its lane, on its direction's communicator; none on the compute stream
    posts = [e for e in log if e[0] == "post"]
    waits = [e for e in log if e[0] == "wait"]
    assert posts and len(posts) == len(waits)
    for e in posts:
        kinds, group, stream = e[1], e[2], e[-1]
        assert stream in ("scatter", "gather") and group == ("grp-in" if stream == "scatter" else "grp-out"), e
    for e in waits:
        assert e[-1] in ("scatter", "gather") and e[1] == ("grp-in" if e[-1] == "scatter" else "grp-out"), e
    scatter_kinds = {k for e in posts if e[-1] == "scatter" for k in e[1]}
    gather_kinds = {k for e in posts if e[-1] == "gather" for k in e[1]}
    assert scatter_kinds == ({"send"} if rank == 0 else {"recv"}) and gather_kinds == ({"recv"} if rank == 0 else {"send"})

    # 2. the compute stream waits for one event per piece (its input) and joins the lanes once, at the very end
    cw = [e for e in log if e[0] == "current_wait_event"]
    assert [e[1] for e in cw] == ["landed %d" % k for k in range(pieces)] and all(e[-1] == "compute" for e in cw)
    joins = [i for i, e in enumerate(log) if e[0] == "current_wait_stream"]
    assert sorted(log[i][1] for i in joins) == ["gather", "scatter"] and min(joins) > max(i for i, e in enumerate(log) if e[0] in ("post", "wait", "encode"))

    # 3. per piece: the order of what is queued
    def idx(pred):
        return [i for i, e in enumerate(log) if pred(e)]
    for k in range(pieces):
        enc_k = idx(lambda e: e[0] == "encode" and e[2] == k)
        rec_k = idx(lambda e: e[0] == "record" and e[1] == "compacted %d" % k)[0]
        assert max(idx(lambda e: e[0] == "compact" and e[2] == k)) < rec_k and log[rec_k][-1] == "compute"
        # the gather lane depends on piece k through that event only
        dep = idx(lambda e: e[0] == "stream_wait_event" and e[1] == "gather" and e[2] == "compacted %d" % k)
        assert len(dep) == 1
        gather_posts_k = [i for i in idx(lambda e: e[0] == "post" and e[-1] == "gather") if i > dep[0]][:2]      # one per codec
        assert len(gather_posts_k) == 2
        if k + 1 < pieces:
            enc_next = idx(lambda e: e[0] == "encode" and e[2] == k + 1)
            # the codecs of piece k + 1 are queued BEFORE the sizes of piece k are exchanged and its gather is posted ...
            totals_k = [i for i in idx(lambda e: e[0] == "host_totals") if i > max(enc_next)][:2]
            assert len(totals_k) == 2 and max(enc_next) < min(totals_k) < dep[0] < min(gather_posts_k)
            # ... and between the two the host waits only for events recorded before those codecs were queued (piece k's packed sizes)
            for i in idx(lambda e: e[0] == "host_wait_event"):
                if max(enc_next) < i < min(gather_posts_k):
                    assert log[i][2] < min(enc_next), log[i]
        assert min(enc_k) > idx(lambda e: e[0] == "current_wait_event" and e[1] == "landed %d" % k)[0]
        if k + 1 < pieces:      # the scatter of piece k + 1 is posted before the codecs of piece k are queued (it overlaps them)
            landed_next = idx(lambda e: e[0] == "record" and e[1] == "landed %d" % (k + 1))[0]
            assert landed_next < min(enc_k) and log[landed_next][-1] == "scatter"

    # 4. nothing but event waits blocks the host, and the sizes never go through a device collective
    assert not [e for e in log if e[0] in ("item", "blocking_copy")]
    assert len(idx(lambda e: e[0] == "host_totals")) == 2 * pieces
