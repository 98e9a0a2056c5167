"""RCCL really executing against the library (north_star: "host code in C calls the kernels ... RCCL broadcast/gather over
xGMI only for the block scatter/gather"): examples/shard_rccl.c -- scatter, FSEHIP_FSE_compress_batch, FSEHIP_compact_batch, the
ncclAllGather of the packed sizes, the grouped variable-length gather, and the packed stream decoded as the check -- run as ONE rank on
the 1-GPU box.  With one rank the root's shard travels root -> root through ncclSend / ncclRecv inside one group, so every RCCL call
of the multi-rank path is issued.  (More ranks: `examples/shard_rccl nBlocks rank world idfile`, one process per GPU.)"""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _exe():
    exe = os.path.join(ROOT, "examples", "shard_rccl")
    if not os.path.exists(exe):
        pytest.skip("examples/shard_rccl not built (make -C examples)")
    return exe


@pytest.mark.parametrize("n_blocks", [1, 777, 4096])
def test_c_host_with_rccl_runs_one_rank(hip, n_blocks):
    p = subprocess.run([_exe(), str(n_blocks)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "shard_rccl OK: %d blocks" % n_blocks in out, out[-2000:]
    assert "through ncclSend / ncclRecv" in out


@pytest.mark.parametrize("n_blocks,pieces", [(4096, 4), (1001, 7), (5, 2)])
def test_c_host_pipelined_protocol_runs_one_rank(hip, n_blocks, pieces):
    """the PIPELINED protocol of DESIGN.md section 5 in C (three streams, a communicator per direction, the compute stream waiting only for its
    next piece, the host only for the event behind the previous piece's compaction): shards in `pieces` pieces, the packed stream gathered piece
    after piece and decoded against the corpus in arrival order"""
    p = subprocess.run([_exe(), str(n_blocks), str(pieces)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "shard_rccl OK: %d blocks" % n_blocks in out and "shard_rccl pipelined OK: %d pieces" % pieces in out, out[-2000:]


def test_c_host_with_rccl_two_ranks_on_one_gpu(hip, tmp_path):
    """two processes sharing the box's one GPU (rank % deviceCount): a real two-rank communicator -- rank 0 sends, rank 1 receives, codes
    and sends back.  RCCL may refuse two ranks on one device; that refusal (not a wrong result) skips the test."""
    idfile = str(tmp_path / "nccl_id")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
    procs = [subprocess.Popen([_exe(), "2001", str(r), "2", idfile, "3"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.skip("two ranks on one GPU did not complete (RCCL does not support it on this box)")
        outs.append(o.decode(errors="replace"))
    if any(p.returncode != 0 for p in procs):
        text = "\n".join(outs)
        if "differ after the round trip" in text or "decode returned" in text:
            pytest.fail(text[-2000:])
        pytest.skip("RCCL refused two ranks on one device: " + text[-300:])
    assert "shard_rccl OK: 2001 blocks" in outs[0], outs[0][-2000:]
