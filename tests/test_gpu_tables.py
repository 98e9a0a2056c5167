"""GPU parity tests of the table-building batch calls (SURVEY 8(a') g1-g3, g5-g6 as calls of their own: FSEHIP_FSE_buildCTable_batch,
FSEHIP_FSE_buildDTable_batch, FSEHIP_HUF_buildCTable_batch, FSEHIP_HUF_readDTableX1_batch) against the reference's builders, table
word for table word, and of the whole using-table pipeline on the device: tables -> *_usingCTable_batch -> *_usingDTable_batch."""
import numpy as np
import pytest
import torch

from oracle.oracle import fse_ctable_u32, fse_dtable_u32, is_error
from test_gpu_fse import mixed_blocks, s64

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle(checker):
    return checker


def _fse_expect(oracle, blk, msv_req, tl_req):
    """what FSE_compress_wksp does up to the table (lib/fse_compress.c:646-665): (result, header bytes, ctable words, tableLog, maxSV)"""
    n = len(blk)
    if n <= 1:
        return 0, None, None, 0, 0
    mx, msv, cnt = oracle.hist_count(blk, msv_req)
    if is_error(mx):
        return mx, None, None, 0, 0
    if mx == n:
        return 1, None, None, 0, 0
    if mx == 1 or mx < (n >> 7):
        return 0, None, None, 0, 0
    tl = oracle.fse_optimal_tablelog(tl_req, n, msv, 2)
    r, norm = oracle.fse_normalize_count(tl, cnt, n, msv)
    if is_error(r):
        return r, None, None, 0, 0
    h, hdr = oracle.fse_write_ncount(512, norm, msv, tl)
    if is_error(h):
        return h, None, None, 0, 0
    _, ct = oracle.fse_build_ctable(norm, msv, tl)
    return h, hdr[:h], ct, tl, msv


@pytest.mark.parametrize("size", [2, 7, 100, 1000, 4097, 32768])
def test_fse_build_ctable_and_dtable_batch(hip, oracle, size):
    for tl_req in (11, 12, 5, 9):
        blocks = mixed_blocks(oracle, 30, size, seed=40 + tl_req)
        src = torch.from_numpy(blocks).cuda()
        ct, hdr, res = hip.fse_build_ctable_batch(src, table_log=tl_req)
        ct_h, hdr_h, res_h = ct.cpu().numpy().view(np.uint32), hdr.cpu().numpy(), res.cpu().numpy()
        built = []
        for b in range(30):
            r, eh, ect, tl, msv = _fse_expect(oracle, blocks[b], 255, tl_req)
            assert res_h[b] == s64(r), (size, tl_req, b, res_h[b], r)
            if eh is None:
                continue
            assert (hdr_h[b][:r] == eh).all(), (size, tl_req, b, "header")
            w = fse_ctable_u32(tl, msv)
            assert (ct_h[b][:w] == ect[:w]).all(), (size, tl_req, b, "ctable", np.nonzero(ct_h[b][:w] != ect[:w])[0][:8])
            built.append(b)
        if not built:
            continue
        idx = torch.tensor(built, device="cuda")
        # DTables from the headers just written (exact header sizes) and from header + trailing bytes
        for pad in (0, 9):
            hsz = res[idx] + pad
            dt, dres = hip.fse_build_dtable_batch(hdr[idx].contiguous(), hsz, max_log=12)
            dt_h, dres_h = dt.cpu().numpy().view(np.uint32), dres.cpu().numpy()
            for i, b in enumerate(built):
                h = int(res_h[b])
                rr, msv, tl, norm = oracle.fse_read_ncount(hdr_h[b][:h + pad])
                assert dres_h[i] == s64(rr), (size, tl_req, b, pad, dres_h[i], rr)
                if is_error(rr):
                    continue
                _, edt = oracle.fse_build_dtable(norm, msv, tl)
                w = fse_dtable_u32(tl)
                assert (dt_h[i][:w] == edt[:w]).all(), (size, tl_req, b, "dtable", np.nonzero(dt_h[i][:w] != edt[:w])[0][:8])
        # maxLog below the table's log: tableLog_tooLarge like FSE_decompress_wksp (lib/fse_decompress.c:266)
        dt, dres = hip.fse_build_dtable_batch(hdr[idx].contiguous(), res[idx], max_log=9)
        for i, b in enumerate(built):
            rr, _, tl, _ = oracle.fse_read_ncount(hdr_h[b][:int(res_h[b])])
            if is_error(rr):             # (a header that ends on a long zero run needs bytes behind it: FSE_readNCount freezes its window near the end)
                assert dres[i].item() == s64(rr), (size, tl_req, b)
            else:
                assert (dres[i].item() == -5) == (tl > 9), (size, tl_req, b)
        # the whole using-table pipeline on the device
        comp, cres = hip.fse_compress_using_ctable_batch(src[idx].contiguous(), ct[idx].contiguous(), max_table_log=12)
        dt, dres = hip.fse_build_dtable_batch(hdr[idx].contiguous(), res[idx], max_log=12)
        good = (cres > 0).nonzero().flatten()
        for b_i in range(len(built)):
            r, out = oracle.fse_compress_using_ctable(blocks[built[b_i]], ct_h[built[b_i]])
            assert cres[b_i].item() == r
        if good.numel():
            out, ores = hip.fse_decompress_using_dtable_batch(comp[good].contiguous(), cres[good].contiguous(), dt[good].contiguous(), size, max_table_log=12)
            comp_h, cres_h, dt_h, out_h, ores_h = comp.cpu().numpy(), cres.cpu().numpy(), dt.cpu().numpy().view(np.uint32), out.cpu().numpy(), ores.cpu().numpy()
            for k, gi in enumerate(good.tolist()):           # (the reference may want more room than the block is long on tiny blocks)
                r, eo = oracle.fse_decompress_using_dtable(comp_h[gi][:cres_h[gi]], dt_h[gi], size)
                assert ores_h[k] == s64(r), (size, tl_req, built[gi], ores_h[k], r)
                if not is_error(r):
                    assert (out_h[k][:r] == eo[:r]).all(), (size, tl_req, built[gi])
                    if size >= 64:
                        assert r == size and (out_h[k][:size] == blocks[built[gi]]).all(), (size, tl_req, built[gi])


def test_fse_build_tables_limits_and_garbage_headers(hip, oracle):
    blocks = mixed_blocks(oracle, 20, 4096, seed=77)
    src = torch.from_numpy(blocks).cuda()
    for msv in (255, 52, 6, 100):
        ct, hdr, res = hip.fse_build_ctable_batch(src, table_log=11, max_symbol_value=msv)
        res_h = res.cpu().numpy()
        for b in range(20):
            r, *_ = _fse_expect(oracle, blocks[b], msv, 11)
            assert res_h[b] == s64(r), (msv, b, res_h[b], r)
    # header capacity too small for the header: FSE_writeNCount's dstSize_tooSmall (lib/fse_compress.c:288-298)
    ct, hdr, res = hip.fse_build_ctable_batch(src, table_log=11, header_capacity=8)
    res_h = res.cpu().numpy()
    for b in range(20):
        n = 4096
        mx, msv, cnt = oracle.hist_count(blocks[b])
        if mx == n or mx == 1 or mx < (n >> 7):
            continue
        tl = oracle.fse_optimal_tablelog(11, n, msv, 2)
        _, norm = oracle.fse_normalize_count(tl, cnt, n, msv)
        h, _ = oracle.fse_write_ncount(8, norm, msv, tl)
        assert res_h[b] == s64(h), (b, res_h[b], h)
    # garbage / truncated headers through the table builder: FSE_readNCount's verdicts
    rng = np.random.default_rng(3)
    cases = []
    blk = oracle.probagen_batch(14, 1, 4096, 9)[0]
    cs, comp = oracle.fse_compress2(blk)
    for t in range(60):
        bad = comp[:cs].copy()
        if t % 3 == 0:
            bad = bad[:int(rng.integers(1, 40))]
        elif t % 3 == 1:
            bad[int(rng.integers(0, 40))] ^= 1 << int(rng.integers(0, 8))
        else:
            bad = rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8)
        cases.append(bad)
    width = max(len(c) for c in cases)
    buf = np.zeros((len(cases), width), np.uint8)
    for i, c in enumerate(cases):
        buf[i, :len(c)] = c
    sizes = torch.tensor([len(c) for c in cases], dtype=torch.int64, device="cuda")
    dt, dres = hip.fse_build_dtable_batch(torch.from_numpy(buf).cuda(), sizes, max_log=12)
    dt_h, dres_h = dt.cpu().numpy().view(np.uint32), dres.cpu().numpy()
    for i, c in enumerate(cases):
        rr, msv, tl, norm = oracle.fse_read_ncount(c)
        if not is_error(rr) and tl > 12:
            rr = (1 << 64) - 5
        assert dres_h[i] == s64(rr), (i, dres_h[i], rr)
        if not is_error(rr):
            _, edt = oracle.fse_build_dtable(norm, msv, tl)
            assert (dt_h[i][:fse_dtable_u32(tl)] == edt).all(), i


@pytest.mark.parametrize("size", [13, 100, 1000, 4097, 32768, 131072])
def test_huf_build_ctable_and_read_dtable_batch(hip, oracle, size):
    from oracle.oracle import huf_compress_bound
    for tl_req in (11, 12, 8):
        blocks = mixed_blocks(oracle, 30, size, seed=60 + tl_req)
        src = torch.from_numpy(blocks).cuda()
        ct, hdr, res = hip.huf_build_ctable_batch(src, table_log=tl_req)
        ct_h, hdr_h, res_h = ct.cpu().numpy().view(np.uint32), hdr.cpu().numpy(), res.cpu().numpy()
        built = []
        for b in range(30):
            blk = blocks[b]
            mx, msv, cnt = oracle.hist_count(blk)
            if mx == size:
                assert res_h[b] == 1 and hdr_h[b][0] == blk[0], (size, b); continue
            if mx <= (size >> 7) + 4:
                assert res_h[b] == 0, (size, b); continue
            hl = oracle.fse_optimal_tablelog(tl_req, size, msv, 1)
            mb, celt = oracle.huf_build_ctable(cnt, msv, hl)
            assert not is_error(mb)
            hs, eh = oracle.huf_write_ctable(256, celt, msv, mb)
            if is_error(hs):
                assert res_h[b] == s64(hs), (size, tl_req, b); continue
            if hs + 12 >= size:
                assert res_h[b] == 0, (size, tl_req, b); continue
            assert res_h[b] == hs, (size, tl_req, b, res_h[b], hs)
            assert (hdr_h[b][:hs] == eh[:hs]).all(), (size, tl_req, b, "header")
            assert (ct_h[b][:msv + 1] == celt[:msv + 1]).all(), (size, tl_req, b, "celt")
            built.append(b)
        if not built:
            continue
        idx = torch.tensor(built, device="cuda")
        for mtl in (11, 12):
            dt, dres = hip.huf_read_dtable_x1_batch(hdr[idx].contiguous(), res[idx], max_table_log=mtl)
            dt_h, dres_h = dt.cpu().numpy().view(np.uint32), dres.cpu().numpy()
            for i, b in enumerate(built):
                h = int(res_h[b])
                rr, edt = oracle.huf_read_dtable_x1(hdr_h[b][:h], mtl)
                assert dres_h[i] == s64(rr), (size, tl_req, b, mtl, dres_h[i], rr)
                if not is_error(rr):
                    tl = (int(edt[0]) >> 16) & 0xFF
                    w = 1 + (1 << (tl - 1))
                    assert (dt_h[i][:w] == edt[:w]).all(), (size, tl_req, b, mtl, hex(dt_h[i][0]), hex(int(edt[0])))
        # the whole using-table pipeline on the device
        sub = src[idx].contiguous()
        comp, cres = hip.huf_compress4x_using_ctable_batch(sub, ct[idx].contiguous())
        dt, dres = hip.huf_read_dtable_x1_batch(hdr[idx].contiguous(), res[idx], max_table_log=12)
        for i, b in enumerate(built):
            r, out = oracle.huf_compress4x_using_ctable(blocks[b], ct_h[b])
            assert cres[i].item() == r, (size, tl_req, b)
        good = ((cres > 0) & (dres > 1)).nonzero().flatten()       # (a tableLog-12 code with a one-bit symbol has a weight HUF_readStats refuses)
        if good.numel():
            out, ores = hip.huf_decompress4x1_using_dtable_batch(comp[good].contiguous(), cres[good].contiguous(), dt[good].contiguous(), size, max_table_log=12)
            assert bool((ores == size).all()) and torch.equal(out, sub[good]), (size, tl_req)


def test_huf_read_dtable_garbage_headers(hip, oracle):
    rng = np.random.default_rng(4)
    cases = []
    for P in (14, 2):
        blk = oracle.probagen_batch(P, 1, 8192, 9)[0]
        cs, comp = oracle.huf_compress2(blk)
        for t in range(40):
            bad = comp[:cs].copy()
            if t % 4 == 0:
                bad = bad[:int(rng.integers(1, 90))]
            elif t % 4 == 1:
                bad[int(rng.integers(0, 30))] ^= 1 << int(rng.integers(0, 8))
            elif t % 4 == 2:
                bad = rng.integers(0, 256, int(rng.integers(1, 140)), dtype=np.uint8)
            cases.append(bad[:200])
    width = max(len(c) for c in cases)
    buf = np.zeros((len(cases), width), np.uint8)
    for i, c in enumerate(cases):
        buf[i, :len(c)] = c
    sizes = torch.tensor([len(c) for c in cases], dtype=torch.int64, device="cuda")
    for mtl in (11, 12):
        dt, dres = hip.huf_read_dtable_x1_batch(torch.from_numpy(buf).cuda(), sizes, max_table_log=mtl)
        dt_h, dres_h = dt.cpu().numpy().view(np.uint32), dres.cpu().numpy()
        for i, c in enumerate(cases):
            rr, edt = oracle.huf_read_dtable_x1(c, mtl)
            assert dres_h[i] == s64(rr), (i, mtl, dres_h[i], rr)
            if not is_error(rr):
                tl = (int(edt[0]) >> 16) & 0xFF
                assert (dt_h[i][:1 + (1 << (tl - 1))] == edt[:1 + (1 << (tl - 1))]).all(), (i, mtl)


def test_huf_zero_filled_dtable_is_refused(hip, oracle):
    """a DTable that was never built (all zero: tableLog 0) must not send the decoders out of bounds: corruption_detected per block
    (the reference's look-up is undefined for tableLog 0: BIT_lookBitsFast would shift by 64)"""
    src = hip.probagen_batch(14, 6, 32768, first_seed=3)
    comp, cres = hip.huf_compress_batch(src)
    hdrs, _ = hip.huf_read_dtable_x1_batch(comp, cres, max_table_log=12)
    zero = torch.zeros_like(hdrs)
    payload = comp[:, 27:].contiguous()
    for fn in (hip.huf_decompress4x1_using_dtable_batch, hip.huf_decompress4x_using_dtable_batch):
        out, res = fn(payload, cres - 27, zero, 32768, max_table_log=12)
        assert (res.cpu().numpy() == -4).all(), res


def test_fse_using_dtable_two_streams_and_stragglers(hip, checker):
    """FSE_decompress_usingDTable over a batch keeps its symbol bytes in a per-workgroup slot of a library scratch (claimed when the workgroup
    starts, handed back when its LAST wave -- service waves and the decoder waves' literal tails -- is through with it).  Two streams
    running the call concurrently over ragged batches (32 KB blocks next to 200-byte ones: workgroup durations differ by orders of
    magnitude, the tails of the short ones run while others claim slots) must not see each other's symbols."""
    n = 6000
    rng = np.random.default_rng(3)
    sizes = np.where(rng.random(n) < 0.5, 32768, rng.integers(200, 2000, n)).astype(np.int64)
    batches = []
    for P, seed in ((14, 1), (80, 50001)):
        src = hip.probagen_batch(P, n, 32768, first_seed=seed)
        d_sizes = torch.from_numpy(sizes).cuda()
        ct, hdr, hres = hip.fse_build_ctable_batch(src, table_log=11, sizes=d_sizes)
        comp, cres = hip.fse_compress_using_ctable_batch(src, ct, max_table_log=11, sizes=d_sizes)
        dt, dres = hip.fse_build_dtable_batch(hdr, hres, max_log=11)
        assert (hres > 1).all() and (cres > 1).all()
        batches.append((src, d_sizes, comp, cres, dt))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    reps = 6
    # destinations made beforehand: nothing between the launches synchronises (the binding's guard check would)
    outs = [[(torch.zeros((n, 32768), dtype=torch.uint8, device="cuda"), torch.zeros(n, dtype=torch.int64, device="cuda")) for _ in range(reps)] for _ in range(2)]
    torch.cuda.synchronize()
    for rep in range(reps):
        for k, (src, d_sizes, comp, cres, dt) in enumerate(batches):
            with torch.cuda.stream(streams[k]):
                hip.fse_decompress_using_dtable_batch(comp, cres, dt, 32768, max_table_log=11, dst=outs[k][rep][0], results=outs[k][rep][1])
    torch.cuda.synchronize()
    col = torch.arange(32768, device="cuda").unsqueeze(0)
    for k, (src, d_sizes, comp, cres, dt) in enumerate(batches):
        mask = col < d_sizes.unsqueeze(1)
        for out, res in outs[k]:
            assert torch.equal(res, d_sizes), "stream %d: regenerated sizes differ" % k
            assert torch.equal(torch.where(mask, out[:, :32768], torch.zeros_like(src)), torch.where(mask, src, torch.zeros_like(src))), "stream %d: bytes differ" % k
