"""The batched calls walk a batch in chunks when the caller's workspace holds fewer blocks than the batch (include/fsehip.h: any
workspace of at least one block's worth is accepted).  Results must not depend on the chunking: every one-shot call is run with the
workspace the library asks for and with one sized for about a third / a seventh of the blocks, and compared bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _small(ws_full, n_blocks, part):
    per = (ws_full.numel() - 2048) // n_blocks
    return torch.empty(2048 + per * max(n_blocks // part, 1), dtype=torch.uint8, device="cuda")


@pytest.mark.parametrize("part", [3, 7])
def test_chunked_workspace_same_results(hip, part):
    n, size = 301, 8192
    src = hip.probagen_mixed((2, 14, 80), n, size)
    # ---- FSE
    ws = hip.fse_workspace(n, 11, False)
    d0, r0 = hip.fse_compress_batch(src, 11, workspace=ws)
    d1, r1 = hip.fse_compress_batch(src, 11, workspace=_small(ws, n, part))
    torch.cuda.synchronize()
    assert torch.equal(r0, r1) and bool((r0 > 1).all())
    for b in range(n):
        assert torch.equal(d0[b, :r0[b]], d1[b, :r1[b]]), b
    wd = hip.fse_workspace(n, 12, True)
    o0, q0 = hip.fse_decompress_batch(d0, r0, size, workspace=wd)
    o1, q1 = hip.fse_decompress_batch(d0, r0, size, workspace=_small(wd, n, part))
    torch.cuda.synchronize()
    assert torch.equal(q0, q1) and torch.equal(o0, o1) and torch.equal(o0, src)
    # ---- Huff0
    ws = hip.huf_workspace(n, False)
    d0, r0 = hip.huf_compress_batch(src, workspace=ws)
    d1, r1 = hip.huf_compress_batch(src, workspace=_small(ws, n, part))
    torch.cuda.synchronize()
    assert torch.equal(r0, r1) and bool((r0 > 1).all())
    for b in range(n):
        assert torch.equal(d0[b, :r0[b]], d1[b, :r1[b]]), b
    wd = hip.huf_workspace(n, True)
    o0, q0 = hip.huf_decompress_batch(d0, r0, size, workspace=wd)
    o1, q1 = hip.huf_decompress_batch(d0, r0, size, workspace=_small(wd, n, part))
    torch.cuda.synchronize()
    assert torch.equal(q0, q1) and torch.equal(o0, o1) and torch.equal(o0, src)
    # ---- 16-bit symbols
    s16 = (src.view(n, size)[:, ::2].to(torch.int16) % 287).contiguous()
    wsz = int(hip.lib.FSEHIP_FSE_compressU16_batch_workspaceSize(n))
    ws = torch.empty(wsz, dtype=torch.uint8, device="cuda")
    d0, r0 = hip.fse_compress_u16_batch(s16, workspace=ws)
    d1, r1 = hip.fse_compress_u16_batch(s16, workspace=_small(ws, n, part))
    torch.cuda.synchronize()
    assert torch.equal(r0, r1)
    ok = torch.nonzero(r0 > 1).flatten()
    assert ok.numel() > n // 2
    for b in ok.cpu().numpy():
        assert torch.equal(d0[b, :r0[b]], d1[b, :r1[b]]), b
    wd = torch.empty(int(hip.lib.FSEHIP_FSE_decompressU16_batch_workspaceSize(n)), dtype=torch.uint8, device="cuda")
    o0, q0 = hip.fse_decompress_u16_batch(d0[ok], r0[ok], s16.shape[1], workspace=wd)
    o1, q1 = hip.fse_decompress_u16_batch(d0[ok], r0[ok], s16.shape[1], workspace=_small(wd, n, part))
    torch.cuda.synchronize()
    assert torch.equal(q0, q1) and torch.equal(o0, o1) and torch.equal(o0, s16[ok])
