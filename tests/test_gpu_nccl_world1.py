"""torch.distributed's NCCL backend (= RCCL on ROCm) initialised for real on the GPU box, at the only world size the box allows: one rank.
What the driver's 8-GPU run does first -- `init_process_group("nccl", device_id=...)`, a second communicator for the way back, the gloo
side group for host integers, barriers and the max-over-ranks reduction of bench.py -- and then the pipelined with-comm job itself
(finitestateentropy_amd.shard.sharded_codec_job_pipelined) on real streams and events with the device codecs: with one rank it posts no
transfer, but every lane, event, pinned size read and join of the multi-rank path runs.  (RCCL moving bytes: tests/test_gpu_rccl.py, the C
host; two ranks of the Python job over gloo on the one GPU: tests/test_gpu_cfg5.py.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import datetime, os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from finitestateentropy_amd import shard
from finitestateentropy_amd.api import FseHip, fse_compress_bound, huf_compress_bound
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%r, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=2))
assert dist.get_backend() == "nccl"
t = torch.ones(1024, device=dev); dist.all_reduce(t); dist.barrier()
assert bool((t == 1).all())
back = dist.new_group()                                   # the communicator of the way back (bench.py makes it the same way)
hg = shard.host_group()                                   # gloo side group for the packed sizes
assert dist.get_backend(hg) == "gloo" and shard.host_group() is hg
assert shard.host_totals(77, 1, hg) == [77]
assert shard.max_over_ranks([1.5, 0.25], dev, 1) == [1.5, 0.25]
hip = FseHip()
n, B = 3001, 32768
corpus = hip.probagen_mixed((2, 14, 80), n, B, first_block=0)

class Codec:
    def __init__(self, name, lo=0, hi=n):
        self.name, self.lo, self.hi, self.src = name, lo, hi, None
    def piece(self, lo, hi):
        return Codec(self.name, lo, hi)
    def encode(self):
        f = hip.fse_compress_batch if self.name == "fse" else hip.huf_compress_batch
        self.dst, self.res = f(self.src)
    def decode(self):
        if self.name == "fse":
            self.out, self.dres = hip.fse_decompress_batch(self.dst, self.res, B, max_log=12)
        else:
            self.out, self.dres = hip.huf_decompress_batch(self.dst, self.res, B)
        assert torch.equal(self.out, self.src) and bool((self.dres == B).all())

for pieces in (1, 4):
    codecs = [Codec("fse"), Codec("huf")]
    mine, gathered, stats = shard.sharded_codec_job_pipelined(corpus, n, B, 0, 1, dev, codecs, lambda pc, src: hip.compact_batch(pc.dst, pc.res, src),
                                                              pieces=pieces, gather_group=back)
    torch.cuda.synchronize()
    assert torch.equal(mine, corpus) and stats["scatter_bytes"] == 0 and stats["order"] == list(range(n))
    for name, (pk, of) in zip(("fse", "huf"), gathered):
        assert int(of[0]) == 0
        if name == "fse":
            out, res = hip.fse_decompress_packed_batch(pk, of, B, B, max_log=12)
        else:
            out, res = hip.huf_decompress_packed_batch(pk, of, B)
        assert bool((res == B).all()) and torch.equal(out, corpus), (name, pieces)
    assert stats["payload_bytes"] == int(gathered[0][1][n]) + int(gathered[1][1][n])
dist.barrier()
dist.destroy_process_group()
print("NCCL_WORLD1_OK", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
'''


def test_nccl_backend_initialises_and_the_pipelined_job_runs_on_real_streams(hip):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = str(s.getsockname()[1]); s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, port)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0 and "NCCL_WORLD1_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
