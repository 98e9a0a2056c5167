"""GPU parity tests for Huff0 (C ABI of libfsehip.so vs the CPU oracle): 4-stream encode, X1 decode, one-shot."""
import numpy as np
import pytest
import torch

from oracle.oracle import huf_compress_bound, is_error
from test_gpu_fse import mixed_blocks, s64

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle(checker):
    """in this module `oracle` is the one-hop checker: the compiled reference itself when oracle/_ref is present"""
    return checker


def _huf_tables(oracle, blk, req):
    mx, msv, cnt = oracle.hist_count(blk)
    n = len(blk)
    if mx == n or msv == 0 or n < 2:
        return None
    hl = oracle.fse_optimal_tablelog(req, n, msv, 1)
    mb, celt = oracle.huf_build_ctable(cnt, msv, hl)
    if is_error(mb):
        return None
    hs, hdr = oracle.huf_write_ctable(256, celt, msv, mb)
    if is_error(hs):
        return None
    r, dt = oracle.huf_read_dtable_x1(hdr[:hs], 11)
    if is_error(r):
        return None
    celt = celt.copy(); celt[msv + 1:] = 0
    return mb, celt, dt


@pytest.mark.parametrize("size", [12, 13, 14, 15, 16, 100, 1001, 4097, 32767, 32768, 65536, 131072])
def test_huf_using_tables_batch(hip, oracle, size):
    for req in (11, 12, 6, 8):
        blocks = mixed_blocks(oracle, 20, size, seed=7 * req)
        keep, cts, dts = [], [], []
        for b in range(20):
            t = _huf_tables(oracle, blocks[b], req)
            if t is None:
                continue
            keep.append(b); cts.append(t[1]); dts.append(t[2])
        if not keep:
            continue
        src = torch.from_numpy(blocks[keep]).cuda()
        d_ct = torch.from_numpy(np.stack(cts).view(np.int32)).cuda()
        d_dt = torch.from_numpy(np.stack(dts).view(np.int32)).cuda()
        full = [oracle.huf_compress4x_using_ctable(blocks[b], cts[i]) for i, b in enumerate(keep)]
        c0 = full[0][0]
        for cap in [huf_compress_bound(size), 0, 16, 17, 18] + ([c0 - 1, c0, c0 + 7, c0 + 8, c0 + 9] if c0 else []):
            if cap < 0:
                continue
            dst, res = hip.huf_compress4x_using_ctable_batch(src, d_ct, dst_capacity=cap)
            dst, res = dst.cpu().numpy(), res.cpu().numpy()
            for i, b in enumerate(keep):
                r, out = oracle.huf_compress4x_using_ctable(blocks[b], cts[i], cap)
                assert res[i] == s64(r), (size, req, cap, b, res[i], r)
                if r:
                    assert (dst[i][:r] == out[:r]).all(), (size, req, cap, b)
        cbuf = np.zeros((len(keep), huf_compress_bound(size)), np.uint8)
        csz = np.zeros(len(keep), np.int64)
        for i, (r, out) in enumerate(full):
            cbuf[i, :r] = out[:r]; csz[i] = r
        ok = csz > 0
        if not ok.any():
            continue
        d_c = torch.from_numpy(cbuf[ok]).cuda(); d_sz = torch.from_numpy(csz[ok]).cuda()
        d_dt_ok = d_dt[torch.from_numpy(ok).cuda()]
        idx = [k for k, f in zip(keep, ok) if f]
        dts_ok = [d for d, f in zip(dts, ok) if f]
        for dsz in (size, size - 1, size + 1, size + 4):
            dst, res = hip.huf_decompress4x1_using_dtable_batch(d_c, d_sz, d_dt_ok, dsz)
            dst, res = dst.cpu().numpy(), res.cpu().numpy()
            for i, b in enumerate(idx):
                r, out = oracle.huf_decompress4x1_using_dtable(cbuf[ok][i][:csz[ok][i]], dts_ok[i], dsz)
                assert res[i] == s64(r), (size, req, dsz, b, res[i], r)
                if dsz == size:
                    assert r == size and (dst[i][:size] == blocks[b]).all(), (size, req, b)


@pytest.mark.parametrize("size", [0, 1, 11, 12, 13, 100, 1000, 4097, 32767, 32768, 131072, 131073])
def test_huf_oneshot_batch(hip, oracle, size):
    for tl in (11, 12, 0, 6, 8):
        blocks = mixed_blocks(oracle, 30, size, seed=31 + tl)
        src = torch.from_numpy(blocks).cuda() if size else torch.zeros((30, 1), dtype=torch.uint8, device="cuda")
        for cap in (huf_compress_bound(size), max(size // 2, 1), 20):
            dst, res = hip.huf_compress_batch(src, table_log=tl, sizes=size, dst_capacity=cap)
            dst, res = dst.cpu().numpy(), res.cpu().numpy()
            if size:
                _, ores, odst = oracle.compress_batch(1, blocks, table_log=tl, cap=cap)
            else:
                ores, odst = np.zeros(30, np.uint64), None
            for b in range(30):
                r = int(ores[b])
                assert res[b] == s64(r), (size, tl, cap, b, res[b], r)
                if not is_error(r) and r >= 1 and odst is not None:
                    assert (dst[b][:r] == odst[b][:r]).all(), (size, tl, cap, b)
            if cap != huf_compress_bound(size) or size == 0:
                continue
            ok = np.array([(not is_error(int(r))) and int(r) > 1 for r in ores])
            if not ok.any():
                continue
            d_c = torch.from_numpy(odst[ok]).cuda(); d_sz = torch.from_numpy(ores[ok].astype(np.int64)).cuda()
            out, dres = hip.huf_decompress_batch(d_c, d_sz, size)
            out, dres = out.cpu().numpy(), dres.cpu().numpy()
            for i, b in enumerate(np.nonzero(ok)[0]):
                r, o = oracle.huf_decompress(odst[b][:int(ores[b])], size)
                assert dres[i] == s64(r), (size, tl, b, dres[i], r)
                if not is_error(r):
                    assert (out[i][:size] == blocks[b]).all(), (size, tl, b)


def test_huf_raw_and_rle_blocks(hip, oracle):
    """HUF_decompress: cSrcSize == dstSize -> stored raw, cSrcSize == 1 -> RLE (lib/huf_decompress.c:1065-1066)."""
    raw = np.arange(100, dtype=np.uint8)
    buf = np.zeros((3, 128), np.uint8)
    buf[0, :100] = raw; buf[1, 0] = 77; buf[2, :5] = 1
    sizes = torch.tensor([100, 1, 200], dtype=torch.int64, device="cuda")
    out, res = hip.huf_decompress_batch(torch.from_numpy(buf).cuda(), sizes, 100)
    res = res.cpu().numpy(); out = out.cpu().numpy()
    assert res[0] == 100 and (out[0][:100] == raw).all()
    assert res[1] == 100 and (out[1][:100] == 77).all()
    assert res[2] == s64(oracle.huf_decompress(buf[2][:128], 100)[0]) or res[2] == -4      # cSrcSize > dstSize


def test_huf_decode_garbage_batch(hip, oracle):
    """programs/fuzzerHuff0.c:228-250 on the device."""
    rng = np.random.default_rng(23)
    cases = []
    for P in (2, 14, 80):
        blk = oracle.probagen_batch(P, 1, 4096, P + 1)[0]
        cs, comp = oracle.huf_compress2(blk)
        comp = comp[:cs]
        for trial in range(100):
            bad = comp.copy()
            kind = trial % 4
            if kind == 0:
                bad = bad[:int(rng.integers(2, cs))]
            elif kind == 1:
                bad[int(rng.integers(0, cs))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 2:
                bad = rng.integers(0, 256, int(rng.integers(2, 300)), dtype=np.uint8)
            else:
                bad[-1] = 0
            cases.append(bad)
    width = max(len(c) for c in cases) + 16
    buf = np.zeros((len(cases), width), np.uint8)
    for i, c in enumerate(cases):
        buf[i, :len(c)] = c
    sizes = np.array([len(c) for c in cases], np.int64)
    d_c = torch.from_numpy(buf).cuda(); d_sz = torch.from_numpy(sizes).cuda()
    for dsz in (4096, 4000, 5000):
        out, res = hip.huf_decompress_batch(d_c, d_sz, dsz)
        res = res.cpu().numpy(); out = out.cpu().numpy()
        for i, c in enumerate(cases):
            r, o = oracle.huf_decompress(c, dsz)
            assert res[i] == s64(r), (i, dsz, len(c), res[i], r)
            if not is_error(r):
                assert (out[i][:r] == o[:r]).all(), (i, dsz)


def test_huf_single_block_host_api(hip, oracle):
    for P, n in ((14, 32768), (80, 4097), (2, 1000), (14, 13)):
        blk = oracle.probagen_batch(P, 1, n, 3)[0]
        a, b = hip.huf_compress2(blk), oracle.huf_compress2(blk)
        assert a[0] == b[0] and (a[1][:a[0]] == b[1][:b[0]]).all()
        if a[0] > 1:
            x = hip.huf_decompress(a[1][:a[0]], n)
            assert x[0] == n and (x[1] == blk).all()
        t = _huf_tables(oracle, blk, 11)
        if t:
            for name in ("huf_compress1x_using_ctable", "huf_compress4x_using_ctable"):
                e1, e2 = getattr(hip, name)(blk, t[1]), getattr(oracle, name)(blk, t[1])
                assert e1[0] == e2[0] and (e1[1][:e1[0]] == e2[1][:e2[0]]).all(), name
            e4 = oracle.huf_compress4x_using_ctable(blk, t[1])
            if e4[0]:
                d1 = hip.huf_decompress4x1_using_dtable(e4[1][:e4[0]], t[2], n)
                assert d1[0] == n and (d1[1] == blk).all()
    rle = np.full(500, 9, np.uint8)
    a, b = hip.huf_compress2(rle), oracle.huf_compress2(rle)
    assert a[0] == b[0] == 1 and a[1][0] == 9


def test_huf_full_size_roundtrip_config4(hip, oracle):
    """BASELINE config 4 shape (Proba14, 32 KB blocks, Huff0 4 streams): bytes == oracle, decode == source;
    checksum of sizes over the first 1000 blocks == 17,266,350 (measured on the reference)."""
    n = 1000
    src = hip.probagen_batch(14, n, 32768, first_seed=1)
    dst, res = hip.huf_compress_batch(src, table_log=11)
    res_h = res.cpu().numpy()
    assert int(res_h.sum()) == 17266350
    host = src.cpu().numpy()
    _, ores, odst = oracle.compress_batch(1, host, table_log=11)
    assert (res_h == ores.astype(np.int64)).all()
    dh = dst.cpu().numpy()
    for b in range(n):
        assert (dh[b][:res_h[b]] == odst[b][:res_h[b]]).all(), b
    out, dres = hip.huf_decompress_batch(dst, res, 32768)
    assert (dres.cpu().numpy() == 32768).all()
    assert torch.equal(out, src)


@pytest.mark.parametrize("seed", [1, 2])
def test_huf_randomized_differential(hip, oracle, seed):
    """One-shot Huff0 compress + decompress on blocks of assorted statistics and sizes: results and bytes identical to the
    CPU oracle; decoding the oracle's streams regenerates the input (results 0 / 1 = stored raw / RLE by the caller)."""
    from test_gpu_fse import _random_blocks
    rng = np.random.default_rng(2000 + seed)
    for size in (300, 2048, int(rng.integers(2049, 40000)), 65536, 131072):
        for hl in (11, int(rng.choice([8, 9, 10, 12]))):
            blocks = _random_blocks(rng, 24, size)
            src = torch.from_numpy(blocks).cuda()
            dst, res = hip.huf_compress_batch(src, table_log=hl)
            dst, res = dst.cpu().numpy(), res.cpu().numpy()
            _, ores, odst = oracle.compress_batch(1, blocks, table_log=hl)
            for b in range(len(blocks)):
                r = int(ores[b])
                assert res[b] == s64(r), (seed, size, hl, b, res[b], r)
                if not is_error(r) and r > 1:
                    assert (dst[b][:r] == odst[b][:r]).all(), (seed, size, hl, b)
            ok = np.array([(not is_error(int(r))) and int(r) > 1 for r in ores])
            if ok.any():
                d_c = torch.from_numpy(odst[ok]).cuda(); d_sz = torch.from_numpy(ores[ok].astype(np.int64)).cuda()
                out, dres = hip.huf_decompress_batch(d_c, d_sz, size)
                assert (dres.cpu().numpy() == size).all(), (seed, size, hl)
                assert (out.cpu().numpy()[:, :size] == blocks[ok]).all(), (seed, size, hl)


def test_huf_x2_tables_accepted(hip, ref):
    """HUF_decompress4X_usingDTable with double-symbol tables built by the reference's HUF_readDTableX2
    (lib/huf_decompress.c:551-649 -> :749-862 via the dispatcher :980-997): batch and single-block calls against the compiled
    reference, valid streams (all distributions, several sizes), truncated / corrupted streams, mixed X1 / X2 tables in one batch;
    the 4X1 entry points keep rejecting X2 tables (huf_decompress.c:411-412)."""
    from oracle.oracle import Oracle
    orc = Oracle()
    rng = np.random.default_rng(8)
    W = 1 + (1 << 12)
    for size in (32768, 5000, 257):
        blocks, tables, streams, kinds = [], [], [], []
        for i, P in enumerate((14, 2, 80, 50, 20, 14, 2, 80)):
            blk = orc.probagen_batch(P, 1, size, 100 + i)[0]
            cs, c = ref.huf_compress2(blk)
            if cs <= 1:
                continue
            x2 = i % 3 != 2                                   # a few X1 tables in the same batch
            h, dt = ref.huf_read_dtable_x2(c[:cs]) if x2 else ref.huf_read_dtable_x1(c[:cs], 11)
            assert not is_error(h)
            blocks.append(blk); tables.append(dt[:W].copy() if len(dt) >= W else np.pad(dt, (0, W - len(dt)))); streams.append(c[h:cs].copy()); kinds.append(x2)
        n = len(blocks)
        cbuf = np.zeros((n, max(len(s) for s in streams) + 8), np.uint8); csz = np.zeros(n, np.int64)
        for i, s in enumerate(streams):
            cbuf[i, :len(s)] = s; csz[i] = len(s)
        d_c = torch.from_numpy(cbuf).cuda(); d_sz = torch.from_numpy(csz).cuda()
        d_dt = torch.from_numpy(np.stack(tables).view(np.int32)).cuda()
        for dst_size in (size, size - 1, size + 3):
            out, res = hip.huf_decompress4x_using_dtable_batch(d_c, d_sz, d_dt, dst_size)
            out, res = out.cpu().numpy(), res.cpu().numpy()
            for i in range(n):
                r, exp = ref.huf_decompress4x_using_dtable(streams[i], tables[i], dst_size)
                assert res[i] == s64(r), (size, dst_size, i, kinds[i], res[i], r)
                if not is_error(r):
                    assert (out[i][:r] == exp[:r]).all(), (size, dst_size, i)
                    if dst_size == size:
                        assert (out[i][:size] == blocks[i]).all()
        # the 4X1 entry point refuses double-symbol tables
        out1, res1 = hip.huf_decompress4x1_using_dtable_batch(d_c, d_sz, d_dt, size)
        res1 = res1.cpu().numpy()
        for i in range(n):
            if kinds[i]:
                assert res1[i] == -1, (i, res1[i])
        # corrupted / truncated streams: same verdicts (and bytes, when the reference accepts them)
        for trial in range(12):
            i = trial % n
            if not kinds[i]:
                continue
            bad = streams[i].copy()
            if trial % 3 == 0:
                bad = bad[:max(10, len(bad) - int(rng.integers(1, 40)))]
            elif trial % 3 == 1:
                pos = rng.integers(0, len(bad), 4); bad[pos] ^= rng.integers(1, 256, 4).astype(np.uint8)
            else:
                bad[:6] = rng.integers(0, 256, 6, dtype=np.uint8)
            r, exp = ref.huf_decompress4x_using_dtable(bad, tables[i], size)
            rg, og = hip.huf_decompress4x_using_dtable(bad, tables[i], size)
            assert rg == r, (size, trial, rg, r)
            if not is_error(r):
                assert (og[:r] == exp[:r]).all()
        # damaged TABLES (cells whose symbols or bit counts no prefix code would produce; the length field stays 1 or 2 so that the
        # reference itself stays inside its buffers): the stream-parallel decoder must not vouch for them -- the literal lock-step
        # decoder reproduces whatever the reference makes of the table, on the valid stream and on a corrupted one
        for trial in range(8):
            i = trial % n
            if not kinds[i]:
                continue
            dt = tables[i].copy()
            tl = (int(dt[0]) >> 16) & 0xFF
            cells = rng.integers(0, 1 << tl, 6)
            for c in cells:
                w = int(dt[1 + c])
                if trial % 4 == 0:
                    w ^= 0x5                                    # another first symbol
                elif trial % 4 == 1:
                    w ^= 0x0300                                 # another second symbol
                elif trial % 4 == 2:
                    w ^= 0x10000                                # bit count off by one
                else:
                    w = (w & 0x00FFFFFF) | ((3 - (w >> 24)) << 24) if (w >> 24) in (1, 2) else w   # one symbol <-> two symbols
                dt[1 + c] = w
            for strm in (streams[i], np.concatenate([streams[i][:6], streams[i][6:][::-1]])):
                r, exp = ref.huf_decompress4x_using_dtable(strm, dt, size)
                rg, og = hip.huf_decompress4x_using_dtable(strm, dt, size)
                assert rg == r, (size, trial, rg, r)
                if not is_error(r):
                    assert (og[:r] == exp[:r]).all(), (size, trial)


@pytest.mark.parametrize("size", [1, 7, 8, 12, 100, 1001, 4097, 32767, 32768, 65536, 131072])
def test_huf_1x_using_ctable_batch(hip, oracle, size):
    """HUF_compress1X_usingCTable over a batch (lib/huf.h:290, lib/huf_compress.c:457-502): bytes and return values against the
    reference for every block, with the capacity sweep of the 4X test (dstSize < 8 -> 0, BIT_closeCStream's overflow rule)."""
    for req in (11, 12, 6):
        blocks = mixed_blocks(oracle, 20, size, seed=5 * req + 1)
        keep, cts = [], []
        for b in range(20):
            t = _huf_tables(oracle, blocks[b], req)
            if t is None:
                continue
            keep.append(b); cts.append(t[1])
        if not keep:
            continue
        src = torch.from_numpy(blocks[keep]).cuda()
        d_ct = torch.from_numpy(np.stack(cts).view(np.int32)).cuda()
        c0 = oracle.huf_compress1x_using_ctable(blocks[keep[0]], cts[0])[0]
        for cap in [huf_compress_bound(size), 0, 7, 8, 9, 16] + ([c0 - 1, c0, c0 + 7, c0 + 8, c0 + 9] if c0 else []):
            if cap < 0:
                continue
            dst, res = hip.huf_compress1x_using_ctable_batch(src, d_ct, dst_capacity=cap)
            dst, res = dst.cpu().numpy(), res.cpu().numpy()
            for i, b in enumerate(keep):
                r, out = oracle.huf_compress1x_using_ctable(blocks[b], cts[i], cap)
                assert res[i] == s64(r), (size, req, cap, b, res[i], r)
                if r:
                    assert (dst[i][:r] == out[:r]).all(), (size, req, cap, b)
        # ragged sizes and one shared table
        if size >= 100:
            sizes = np.array([max(1, size - 37 * i) for i in range(len(keep))], np.int64)
            dst, res = hip.huf_compress1x_using_ctable_batch(src, d_ct, sizes=torch.from_numpy(sizes).cuda())
            dst, res = dst.cpu().numpy(), res.cpu().numpy()
            for i, b in enumerate(keep):
                r, out = oracle.huf_compress1x_using_ctable(blocks[b][:sizes[i]], cts[i], huf_compress_bound(size))
                assert res[i] == s64(r) and (dst[i][:r] == out[:r]).all(), (size, req, b)


def test_huf_1x_full_size(hip, oracle):
    """north_star names HUF_compress1X_usingCTable: 20k x 32 KB Proba14 blocks with one shared table; every block against the
    reference on a strided sample, all sizes plausible"""
    n = 20000
    src = hip.probagen_batch(14, n, 32768, first_seed=1)
    host0 = src[0].cpu().numpy()
    t = _huf_tables(oracle, host0, 11)
    d_ct = torch.from_numpy(t[1].view(np.int32)).cuda().view(1, -1)
    dst, res = hip.huf_compress1x_using_ctable_batch(src, d_ct, shared_table=True)
    assert int((res > 16000).sum()) == n and int((res < 18500).sum()) == n
    idx = np.arange(0, n, n // 64)
    hs, hd, hr = src[idx].cpu().numpy(), dst[idx].cpu().numpy(), res[idx].cpu().numpy()
    for k in range(len(idx)):
        r, out = oracle.huf_compress1x_using_ctable(hs[k], t[1])
        assert hr[k] == r and (hd[k][:r] == out[:r]).all(), int(idx[k])


def test_oneshot_batch_argument_errors(hip, oracle):
    """INTEGRATION: bad one-shot arguments are PER-BLOCK results, exactly what FSE_compress2 / HUF_compress2 return for that block
    (lib/fse_compress.c:691; lib/huf_compress.c:654-660 in the reference's order of checks)"""
    blocks = mixed_blocks(oracle, 12, 4096, seed=3)
    src = torch.from_numpy(blocks).cuda()
    sizes = torch.tensor([4096, 0, 4096, 1, 2, 4096, 100, 4096, 0, 4096, 4096, 7], dtype=torch.int64, device="cuda")
    hs = sizes.cpu().numpy()
    for tl, msv in ((13, 255), (14, 100), (12, 256), (12, 300), (11, 256), (9, 700), (11, 255)):     # (within what the reference defines)
        _, res = hip.fse_compress_batch(src, table_log=tl, max_symbol_value=msv, sizes=sizes)
        res = res.cpu().numpy()
        for b in range(12):
            r, _ = oracle.fse_compress2(blocks[b][:hs[b]], msv, tl)
            assert res[b] == s64(r), ("fse", tl, msv, b, res[b], r)
    for tl, msv in ((13, 255), (12, 256), (13, 256), (14, 300), (11, 255)):
        for cap in (huf_compress_bound(4096), 0):
            _, res = hip.huf_compress_batch(src, table_log=tl, max_symbol_value=msv, sizes=sizes, dst_capacity=cap)
            res = res.cpu().numpy()
            for b in range(12):
                r, _ = oracle.huf_compress2(blocks[b][:hs[b]], msv, tl, cap)
                assert res[b] == s64(r), ("huf", tl, msv, cap, b, res[b], r)
    # a block beyond HUF_BLOCKSIZE_MAX meets srcSize_wrong before the table-log check
    big = torch.zeros((2, 131073), dtype=torch.uint8, device="cuda")
    _, res = hip.huf_compress_batch(big, table_log=13)
    r, _ = oracle.huf_compress2(np.zeros(131073, np.uint8), 255, 13)
    assert res.cpu().numpy()[0] == s64(r)


def test_huf_oneshot_valid_max_symbol_value_below_255(hip, oracle):
    """HUF_compress2 with a VALID maxSymbolValue < 255 (lib/huf_compress.c:661-671 -> HIST_count_wksp, lib/hist.c:128,169-170):
    exact fit, one too small (maxSymbolValue_tooSmall per block), generous limits; bytes, return values, decode."""
    n = 48
    for P in (14, 80, 2):
        src = hip.probagen_batch(P, n, 32768, first_seed=91)
        host = src.cpu().numpy()
        seen = int(host.max())
        for msv in sorted({seen, seen - 1, seen + 1, max(seen // 2, 1), 1, 100, 200, 254} - {0}):
            if msv > 255:
                continue
            for tl in (11, 12, 0):
                dst, res = hip.huf_compress_batch(src, table_log=tl, max_symbol_value=msv)
                dh, rh = dst.cpu().numpy(), res.cpu().numpy()
                for b in range(0, n, 5):
                    r, out = oracle.huf_compress2(host[b], msv, tl)
                    assert rh[b] == s64(r), (P, msv, tl, b, rh[b], r)
                    if msv < int(host[b].max()):
                        assert rh[b] == -7
                    elif r > 1:
                        assert (dh[b][:r] == out[:r]).all(), (P, msv, tl, b)
                if msv >= seen and (rh > 1).all():
                    out, dres = hip.huf_decompress_batch(dst, res, 32768)
                    assert (dres.cpu().numpy() == 32768).all() and torch.equal(out, src), (P, msv, tl)
    blocks = mixed_blocks(oracle, 30, 4096, seed=19)
    src = torch.from_numpy(blocks).cuda()
    sizes = torch.tensor([4096 - 41 * (b % 5) for b in range(30)], dtype=torch.int64, device="cuda")
    hs = sizes.cpu().numpy()
    for msv in (6, 52, 53, 199, 200, 254):
        dst, res = hip.huf_compress_batch(src, table_log=11, max_symbol_value=msv, sizes=sizes)
        dh, rh = dst.cpu().numpy(), res.cpu().numpy()
        for b in range(30):
            r, out = oracle.huf_compress2(blocks[b][:hs[b]], msv, 11)
            assert rh[b] == s64(r), (msv, b, rh[b], r)
            if not is_error(r) and r > 1:
                assert (dh[b][:r] == out[:r]).all(), (msv, b)


@pytest.mark.timeout(300)
def test_huf_x1_damaged_tables(hip, ref):
    """Caller-built SINGLE-symbol tables that no HUF_readDTableX1 would produce -- cells with nbBits 0 (a zero-filled or half-built
    table), bit counts beyond the table log, other symbols -- through the batch call and the single-block call: the stream-parallel
    decoder advances by the cells' nbBits alone and must not vouch for (or spin on) such a table; whatever the reference makes of it
    (lib/huf_decompress.c:214-237,262-354: its loops are bounded by the output pointer) is reproduced by the serial kernel."""
    from oracle.oracle import Oracle
    orc = Oracle()
    rng = np.random.default_rng(21)
    W = 1 + (1 << 11)
    for size in (32768, 6000):
        blocks, tables, streams = [], [], []
        for i, P in enumerate((14, 2, 80, 50, 14, 2)):
            blk = orc.probagen_batch(P, 1, size, 300 + i)[0]
            cs, c = ref.huf_compress2(blk)
            assert cs > 1
            h, dt = ref.huf_read_dtable_x1(c[:cs], 11)
            assert not is_error(h)
            dt = np.pad(dt, (0, max(W - len(dt), 0)))[:W].copy()
            tl = (int(dt[0]) >> 16) & 0xFF
            cells = dt[1:].view(np.uint16)                        # {byte, nbBits} per cell, low byte first
            kind = i % 6
            pick = rng.integers(0, 1 << tl, 5)
            if kind == 0:
                cells[pick] &= 0x00FF                             # nbBits 0 in a few cells
            elif kind == 1:
                cells[: 1 << tl] = 0                              # zero-filled table behind a plausible descriptor
            elif kind == 2:
                cells[pick] = (cells[pick] & 0x00FF) | ((tl + 1 + (pick & 3)).astype(np.uint16) << 8)   # more bits than the table log
            elif kind == 3:
                cells[pick] ^= 0x0011                             # other symbols (a valid decode of other bytes)
            elif kind == 4:
                cells[0] &= 0x00FF                                # the all-zero index only
            else:
                cells[(1 << tl) - 1] = 0xFF00 | (cells[(1 << tl) - 1] & 0xFF)   # nbBits 255 on the all-ones index
            blocks.append(blk); tables.append(dt); streams.append(c[h:cs].copy())
        n = len(blocks)
        cbuf = np.zeros((n, max(len(s) for s in streams) + 8), np.uint8); csz = np.zeros(n, np.int64)
        for i, s in enumerate(streams):
            cbuf[i, :len(s)] = s; csz[i] = len(s)
        d_c = torch.from_numpy(cbuf).cuda(); d_sz = torch.from_numpy(csz).cuda()
        d_dt = torch.from_numpy(np.stack(tables).view(np.int32)).cuda()
        for fn in (hip.huf_decompress4x1_using_dtable_batch, hip.huf_decompress4x_using_dtable_batch):
            out, res = fn(d_c, d_sz, d_dt, size, max_table_log=11)
            out, res = out.cpu().numpy(), res.cpu().numpy()
            for i in range(n):
                r, exp = ref.huf_decompress4x1_using_dtable(streams[i], tables[i], size)
                assert res[i] == s64(r), (size, i, res[i], r)
                if not is_error(r):
                    assert (out[i][:r] == exp[:r]).all(), (size, i)
        for i in range(n):
            r, exp = ref.huf_decompress4x1_using_dtable(streams[i], tables[i], size)
            rg, og = hip.huf_decompress4x1_using_dtable(streams[i], tables[i], size)
            assert rg == r, (size, i, rg, r)
            if not is_error(r):
                assert (og[:r] == exp[:r]).all(), (size, i)


@pytest.mark.parametrize("size", [1, 8, 100, 1001, 4097, 32767, 32768, 65536, 131072])
def test_huf_1x_decode_using_dtable(hip, ref, size):
    """HUF_decompress1X1_usingDTable / HUF_decompress1X_usingDTable (lib/huf.h:318-320, lib/huf_decompress.c:239-260,724-747,961-975) over a
    batch and on host pointers: single-stream blocks written by HUF_compress1X_usingCTable, single- and double-symbol tables, several
    destination sizes, truncated and corrupted streams -- results and bytes against the compiled reference.  Streams of 32 KB blocks
    exceed the stream-parallel decoder's LDS budget and are decoded in pieces."""
    from oracle.oracle import Oracle
    orc = Oracle()
    rng = np.random.default_rng(size)
    W = 1 + (1 << 12)
    blocks, t1, t2, streams = [], [], [], []
    for i, P in enumerate((14, 2, 80, 50, 20, 14, 2, 80, 14, 5)):
        blk = orc.probagen_batch(P, 1, max(size, 64), 500 + i)[0][:size] if size >= 1 else np.zeros(0, np.uint8)
        big = orc.probagen_batch(P, 1, 32768, 500 + i)[0]              # the table comes from a full block of the distribution
        cs, c = ref.huf_compress2(big)
        assert cs > 1
        h, x1 = ref.huf_read_dtable_x1(c[:cs], 11)
        h2, x2 = ref.huf_read_dtable_x2(c[:cs])
        assert not is_error(h) and not is_error(h2)
        # the matching code table, to write the stream with the reference's HUF_compress1X_usingCTable
        mx, msv, cnt = ref.hist_count(big)
        hl = ref.fse_optimal_tablelog(11, 32768, msv, 1)
        mb, celt = ref.huf_build_ctable(cnt, msv, hl)
        if int(blk.max(initial=0)) > msv:
            continue
        r, out = ref.huf_compress1x_using_ctable(blk, celt)
        if r == 0:
            continue
        blocks.append(blk); streams.append(out[:r].copy())
        t1.append(np.pad(x1, (0, max(W - len(x1), 0)))[:W].copy()); t2.append(np.pad(x2, (0, max(W - len(x2), 0)))[:W].copy())
    if not blocks:
        pytest.skip("nothing compressible at this size")
    n = len(blocks)
    cbuf = np.zeros((n, max(len(s) for s in streams) + 8), np.uint8); csz = np.zeros(n, np.int64)
    for i, s in enumerate(streams):
        cbuf[i, :len(s)] = s; csz[i] = len(s)
    d_c = torch.from_numpy(cbuf).cuda(); d_sz = torch.from_numpy(csz).cuda()
    for tabs, kind in ((t1, "x1"), (t2, "x2")):
        d_dt = torch.from_numpy(np.stack(tabs).view(np.int32)).cuda()
        for dst_size in sorted({size, max(size - 1, 1), size + 3}):
            for fn_name in ("huf_decompress1x1_using_dtable_batch", "huf_decompress1x_using_dtable_batch"):
                out, res = getattr(hip, fn_name)(d_c, d_sz, d_dt, dst_size)
                out, res = out.cpu().numpy(), res.cpu().numpy()
                for i in range(n):
                    if kind == "x2" and "1x1" in fn_name:
                        assert res[i] == -1, (size, i, res[i]); continue         # lib/huf_decompress.c:367-369
                    r, exp = ref.huf_decompress1x_using_dtable(streams[i], tabs[i], dst_size)
                    assert res[i] == s64(r), (size, kind, dst_size, fn_name, i, res[i], r)
                    if not is_error(r):
                        assert (out[i][:r] == exp[:r]).all(), (size, kind, dst_size, i)
                        if dst_size == size:
                            assert (out[i][:size] == blocks[i]).all()
        # damaged streams through the single-block calls
        for trial in range(9):
            i = trial % n
            bad = streams[i].copy()
            if len(bad) < 3:
                continue
            if trial % 3 == 0:
                bad = bad[:max(1, len(bad) - int(rng.integers(1, 20)))]
            elif trial % 3 == 1:
                pos = rng.integers(0, len(bad), 3); bad[pos] ^= rng.integers(1, 256, 3).astype(np.uint8)
            else:
                bad[-1] = 0
            r, exp = ref.huf_decompress1x_using_dtable(bad, tabs[i], size)
            rg, og = hip.huf_decompress1x_using_dtable(bad, tabs[i], size)
            assert rg == r, (size, kind, trial, rg, r)
            if not is_error(r):
                assert (og[:r] == exp[:r]).all(), (size, kind, trial)


def test_huf_1x_roundtrip_on_the_device_full_size(hip, oracle):
    """the library reads what its own 1X encoder writes: 2,000 x 32 KB blocks, tables built on the device, HUF_compress1X_usingCTable_batch ->
    HUF_decompress1X1_usingDTable_batch; and 64 KB / 128 KB blocks through the 4-stream calls (streams beyond the LDS budget: pieces)"""
    src = hip.probagen_mixed((2, 14, 80), 2000)
    ct, hdr, hres = hip.huf_build_ctable_batch(src)
    dt, dres = hip.huf_read_dtable_x1_batch(hdr, hres, max_table_log=11)
    assert bool((hres > 1).all()) and bool((dres == hres).all())
    comp, cres = hip.huf_compress1x_using_ctable_batch(src, ct)
    assert bool((cres > 0).all())
    out, res = hip.huf_decompress1x1_using_dtable_batch(comp, cres, dt, 32768, max_table_log=11)
    assert bool((res == 32768).all()) and torch.equal(out, src)
    for size in (65536, 131072):
        big = hip.probagen_mixed((14, 80, 2), 96, block_size=size)
        c4, r4 = hip.huf_compress_batch(big)
        o4, q4 = hip.huf_decompress_batch(c4, r4, size)
        assert bool((q4 == size).all()) and torch.equal(o4, big), size
        _, ores, odst = oracle.compress_batch(1, big.cpu().numpy()[:8])
        assert (r4.cpu().numpy()[:8] == ores.astype(np.int64)).all()


def test_huf_compress_barely_compressible_blocks_in_two_halves(hip, oracle):
    """Blocks that compress by 0.5 ... 6 %: their four streams total more than the encoder's 30 KiB LDS image (five workgroups per CU since
    round 6), so k_huf_encode emits them in two halves (jump table + streams 0, 1, then streams 2, 3); below that the one-pass path, above it
    "not compressible" -- sizes, bytes and round trip against the reference across the whole band, whole and ragged blocks, odd destinations"""
    rng = np.random.default_rng(606)
    n, size = 96, 32768
    blocks = np.zeros((n, size), np.uint8)
    for b in range(n):
        extra = 0.002 + 0.0016 * b                                            # share of the block that is one favoured byte: 0.2 ... 15 %
        x = rng.integers(0, 256, size, dtype=np.uint8)
        k = int(extra * size)
        x[rng.choice(size, k, replace=False)] = 77
        blocks[b] = x
    src = torch.from_numpy(blocks).cuda()
    sizes = torch.full((n,), size, dtype=torch.int64, device="cuda")
    sizes[1::3] = torch.from_numpy(rng.integers(20000, size, len(range(1, n, 3)))).cuda()
    dst, res = hip.huf_compress_batch(src, table_log=11, sizes=sizes)
    res_h, dst_h, sz_h = res.cpu().numpy(), dst.cpu().numpy(), sizes.cpu().numpy()
    band = 0
    for b in range(n):
        r, out = oracle.huf_compress2(blocks[b][:sz_h[b]], 255, 11)
        assert res_h[b] == s64(r), (b, sz_h[b], res_h[b], r)
        if r > 1:
            assert (dst_h[b][:r] == out[:r]).all(), (b, sz_h[b], r)
            band += r > 30720
    assert band >= 8, band                                                     # the two-halves path was really taken
    coded = res > 1
    out, dres = hip.huf_decompress_batch(dst[coded].contiguous(), res[coded].contiguous(), sizes[coded].contiguous())
    out_h, dres_h = out.cpu().numpy(), dres.cpu().numpy()
    for i, b in enumerate(np.nonzero(coded.cpu().numpy())[0]):
        assert dres_h[i] == sz_h[b] and (out_h[i][:sz_h[b]] == blocks[b][:sz_h[b]]).all(), b


def test_huf_advanced_flow_end_to_end_on_the_device(hip, ref):
    """count -> HUF_buildCTable -> HUF_writeCTable -> HUF_compress4X / 1X_usingCTable | HUF_readDTableX1 -> HUF_decompress4X1 / 1X1_usingDTable, and the
    one-call forms HUF_compress1X / HUF_decompress4X1 / HUF_decompress1X1: every step a device call under the reference's name and signature
    (include/fsehip.h), every intermediate against the reference run on the same inputs"""
    rng = np.random.default_rng(31)
    for size, p in ((32768, 0.14), (32768, 0.8), (5000, 0.05), (131072, 0.3)):
        src = np.minimum(rng.geometric(p, size) - 1, 255).astype(np.uint8)
        mx, msv, count = hip.hist_count(src, 255)
        log, celt = hip.huf_build_ctable(count, msv, 0)
        rlog, rcelt = ref.huf_build_ctable(count, msv, 0)
        assert log == rlog and (celt[:msv + 1] == rcelt[:msv + 1]).all(), (size, p)
        h, hdr = hip.huf_write_ctable(celt, msv, log)
        rh, rhdr = ref.huf_write_ctable(256, rcelt, msv, rlog)
        assert h == rh and (hdr[:h] == rhdr[:h]).all(), (size, p)
        for streams in (4, 1):
            enc = hip.huf_compress4x_using_ctable if streams == 4 else hip.huf_compress1x_using_ctable
            renc = ref.huf_compress4x_using_ctable if streams == 4 else ref.huf_compress1x_using_ctable
            c, comp = enc(src, celt)
            rc, rcomp = renc(src, rcelt)
            assert c == rc and c > 0 and (comp[:c] == rcomp[:c]).all(), (size, p, streams)
            block = np.concatenate([hdr[:h], comp[:c]])
            g, dt = hip.huf_read_dtable_x1(block, 12)
            assert g == h
            dec = hip.huf_decompress4x1_using_dtable if streams == 4 else hip.huf_decompress1x1_using_dtable
            d, out = dec(block[g:], dt, size)
            assert d == size and (out[:size] == src).all(), (size, p, streams)
            one = hip.huf_decompress4x1 if streams == 4 else hip.huf_decompress1x1
            d, out = one(block, size)
            assert d == size and (out[:size] == src).all(), (size, p, streams, "one call")
        c1, comp1 = hip.huf_compress1x(src, 255, 11)
        d, out = hip.huf_decompress1x1(comp1[:c1], size)
        assert c1 > 1 and d == size and (out[:size] == src).all(), (size, p, "HUF_compress1X")
