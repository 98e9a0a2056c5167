"""The batched entry points are pure stream work (kernel launches on the caller's stream and nothing else, device-side
lists instead of host read-backs), so a caller with a launch-bound inner loop -- small batches, many calls -- can capture them in a
HIP graph and replay it (include/fsehip.h "Streams and graphs").  Captured here through torch.cuda.graph (hipStreamBeginCapture on
torch's capture stream, which api.py hands to the library): the replays must produce what direct calls produce, for new source
bytes in the same buffers, on both codecs and both directions."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _valid(dst, res):
    k = int(res.max())
    m = torch.arange(k, device=dst.device)[None, :] < res[:, None]
    return dst[:, :k] * m


@pytest.mark.parametrize("codec", ["fse", "huf"])
def test_batched_calls_replay_from_a_hip_graph(hip, checker, codec):
    n, size = 96, 32768
    comp = hip.fse_compress_batch if codec == "fse" else hip.huf_compress_batch
    deco = (lambda c, r: hip.fse_decompress_batch(c, r, size)) if codec == "fse" else (lambda c, r: hip.huf_decompress_batch(c, r, size))
    src = hip.probagen_batch(14, n, size, first_seed=11).clone()
    # static buffers of the graph
    cws = hip.fse_workspace(n, 11, False) if codec == "fse" else hip.huf_workspace(n, False)
    dws = hip.fse_workspace(n, 12, True) if codec == "fse" else hip.huf_workspace(n, True)
    cdst, cres = comp(src, workspace=cws)                                   # warm-up outside the capture (one-time function attributes)
    out, dres = deco(cdst, cres) if codec == "huf" else hip.fse_decompress_batch(cdst, cres, size, workspace=dws)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        if codec == "fse":
            hip.fse_compress_batch(src, dst=cdst, results=cres, workspace=cws)
            hip.fse_decompress_batch(cdst, cres, size, dst=out, results=dres, workspace=dws)
        else:
            hip.huf_compress_batch(src, dst=cdst, results=cres, workspace=cws)
            hip.huf_decompress_batch(cdst, cres, size, dst=out, results=dres, workspace=dws)
    for trial, (proba, seed) in enumerate(((14, 500), (80, 900), (2, 1300))):
        src.copy_(hip.probagen_batch(proba, n, size, first_seed=seed))
        cdst.zero_(); out.zero_()
        g.replay()
        torch.cuda.synchronize()
        ed, er = comp(src)                                                   # the same work by direct calls into fresh buffers
        assert torch.equal(cres, er), (codec, trial)
        assert torch.equal(_valid(cdst, cres), _valid(ed, er)), (codec, trial)
        assert bool((dres == size).all()) and torch.equal(out[:, :size], src), (codec, trial)
        # and against the reference itself on a few blocks
        h = src[:4].cpu().numpy()
        for i in range(4):
            r, ref_bytes = (checker.fse_compress2(h[i]) if codec == "fse" else checker.huf_compress2(h[i]))
            assert int(cres[i]) == r and (cdst[i, :r].cpu().numpy() == ref_bytes[:r]).all(), (codec, trial, i)
    # what the capture buys a launch-bound caller: time per round trip of this small batch, direct calls against one graph launch
    def timed(fn, reps=30):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    def direct():
        if codec == "fse":
            hip.fse_compress_batch(src, dst=cdst, results=cres, workspace=cws)
            hip.fse_decompress_batch(cdst, cres, size, dst=out, results=dres, workspace=dws)
        else:
            hip.huf_compress_batch(src, dst=cdst, results=cres, workspace=cws)
            hip.huf_decompress_batch(cdst, cres, size, dst=out, results=dres, workspace=dws)
    td, tg = timed(direct), timed(g.replay)
    print("\n%s round trip of %d x 32 KB blocks: direct calls %.1f us, graph replay %.1f us" % (codec, n, td * 1e6, tg * 1e6))
    assert tg < 2.0 * td + 1e-3                                              # (not a performance gate: only that the replay is sane)


def test_u16_batched_calls_replay_from_a_hip_graph(hip):
    """the 16-bit-symbol coder's batch calls (two table-builder launches, the LDS decoder, the lane-per-block kernels) captured too"""
    import ctypes as C
    n, nsym = 48, 16384
    g0 = torch.Generator(device="cuda"); g0.manual_seed(5)

    def make(seed):
        g0.manual_seed(seed)
        u = torch.rand((n, nsym), device="cuda", generator=g0)
        return (torch.log1p(-u) / torch.log(torch.tensor(0.92, device="cuda"))).clamp_(0, 286).to(torch.int16)   # geometric over 287 symbols
    src = make(1)
    SZ = C.c_size_t
    cws = torch.empty(int(hip.lib.FSEHIP_FSE_compressU16_batch_workspaceSize(SZ(n))), dtype=torch.uint8, device="cuda")
    dws = torch.empty(int(hip.lib.FSEHIP_FSE_decompressU16_batch_workspaceSize(SZ(n))), dtype=torch.uint8, device="cuda")
    cdst, cres = hip.fse_compress_u16_batch(src, workspace=cws)
    out, dres = hip.fse_decompress_u16_batch(cdst, cres, nsym, workspace=dws)
    torch.cuda.synchronize()
    assert bool((dres == nsym).all()) and torch.equal(out[:, :nsym], src)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        hip.fse_compress_u16_batch(src, dst=cdst, results=cres, workspace=cws)
        hip.fse_decompress_u16_batch(cdst, cres, nsym, dst=out, results=dres, workspace=dws)
    for seed in (2, 3, 4, 5):
        src.copy_(make(seed))
        cdst.zero_(); out.zero_()
        g.replay()
        torch.cuda.synchronize()
        ed, er = hip.fse_compress_u16_batch(src)
        assert torch.equal(cres, er) and torch.equal(_valid(cdst, cres), _valid(ed, er)), seed
        assert bool((dres == nsym).all()) and torch.equal(out[:, :nsym], src), seed
        _ = src[:2].cpu()                                                     # (a host synchronisation between the replays)


def test_using_table_and_packed_calls_replay_from_a_hip_graph(hip):
    """the round-4 calls captured too: tables built on the device, the using-table hot loops (the caller-table FSE decoder claims a slot of the
    library's symbol scratch per workgroup: allocated by the warm-up call, before the capture), compaction, the packed decoder"""
    from finitestateentropy_amd.api import FseHip, fse_compress_bound
    n, size = 96, 32768
    old = FseHip.guard
    FseHip.guard = 0                                      # (the guard's check synchronises: not inside a capture)
    try:
        src = hip.probagen_batch(14, n, size, first_seed=21).clone()
        cap = fse_compress_bound(size)
        comp = torch.empty((n, cap), dtype=torch.uint8, device="cuda"); cres = torch.empty(n, dtype=torch.int64, device="cuda")
        out = torch.empty((n, size), dtype=torch.uint8, device="cuda"); ores = torch.empty(n, dtype=torch.int64, device="cuda")
        out2 = torch.empty((n, size), dtype=torch.uint8, device="cuda"); ores2 = torch.empty(n, dtype=torch.int64, device="cuda")
        packed = torch.empty(n * size, dtype=torch.uint8, device="cuda"); offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        dws = hip.fse_workspace(n, 12, True)

        def work():
            ct, hdr, hres = hip.fse_build_ctable_batch(src, table_log=11)
            hip.fse_compress_using_ctable_batch(src, ct, max_table_log=11, dst=comp, results=cres)
            dt, _ = hip.fse_build_dtable_batch(hdr, hres, max_log=11)
            hip.fse_decompress_using_dtable_batch(comp, cres, dt, size, max_table_log=12, dst=out, results=ores)
            hip.compact_batch(comp, cres, src, packed=packed, offsets=offsets)
            return hdr, hres
        work(); torch.cuda.synchronize()                  # warm-up outside the capture
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            hdr, hres = work()
        for trial, (proba, seed) in enumerate(((14, 510), (80, 910), (2, 1310), (14, 77))):
            src.copy_(hip.probagen_batch(proba, n, size, first_seed=seed))
            comp.zero_(); out.zero_()
            g.replay()
            torch.cuda.synchronize()
            ok = cres > 0
            assert bool((ores[ok] == size).all()) and torch.equal(out[ok], src[ok]), trial
            ed, er = hip.fse_compress_batch(src, table_log=11)   # header || payload = the one-shot block
            assert bool(((hres + cres == er) | ~ok | (er <= 1)).all()), trial
            assert int(offsets[n].item()) == int(torch.where(cres > 1, cres, torch.where(cres == 0, torch.full_like(cres, size), cres)).sum().item()), trial
            _ = src[:2].cpu()
    finally:
        FseHip.guard = old
