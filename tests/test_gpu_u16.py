"""GPU parity tests of the 16-bit-symbol coder (lib/fseU16.c) through the C ABI against the COMPILED
REFERENCE (oracle/_ref/libfse_ref.so, which travels with the snapshot): FSE_countU16, FSE_compressU16, FSE_decompressU16 --
compressed bytes, return values (0 / 1 / sizes / error codes), regenerated symbols; valid, truncated and corrupted streams;
the batched calls with ragged sizes.  Cases in which the reference itself is undefined (room of 8 bytes or less behind the
header, a stream without payload: include/fsehip.h) are asserted against the documented device behaviour instead."""
import numpy as np
import pytest
import torch

from oracle.oracle import Ref, is_error

pytestmark = pytest.mark.gpu
MAXSV = 286


@pytest.fixture(scope="module")
def ref():
    if not Ref.available():
        pytest.skip("oracle/_ref/libfse_ref.so not built (needs the reference tree: make -C oracle ref)")
    return Ref()


def s64(v):
    v = int(v)
    return v - (1 << 64) if v >= (1 << 63) else v


def u16_block(rng, n, kind):
    """n symbols of a 16-bit alphabet (all <= MAXSV): the fuzzer's two generators (programs/fuzzerU16.c:107-134) and some others"""
    if kind in ("p8", "p80"):
        p, start = (0.08, 240) if kind == "p8" else (0.80, 257)
        table = np.zeros(4096, np.uint16)
        remaining, pos, val = 4096, 0, start
        while remaining:
            k = int(remaining * p) + 1
            table[pos:pos + k] = val
            pos += k; remaining -= k
            val += 1
            if val >= MAXSV:
                val = 1
        return table[rng.integers(0, 4096, n)]
    if kind == "flat":
        return rng.integers(0, MAXSV + 1, n).astype(np.uint16)
    if kind == "small":
        return rng.integers(0, 5, n).astype(np.uint16) * 70
    if kind == "geo":
        return np.minimum(rng.geometric(0.05, n) - 1, MAXSV).astype(np.uint16)
    if kind == "rle":
        return np.full(n, 123, np.uint16)
    if kind == "rare":                                   # one dominant symbol, a few stragglers: counters of -1, slow mixing
        a = np.full(n, 7, np.uint16)
        if n:
            idx = rng.integers(0, n, max(n // 300, 1))
            a[idx] = rng.integers(0, MAXSV + 1, idx.size)
        return a
    raise ValueError(kind)


KINDS = ("p8", "p80", "flat", "small", "geo", "rare")


def test_count_u16(hip, ref):
    rng = np.random.default_rng(1)
    for n in (0, 1, 2, 100, 5000, 40000):
        for kind in KINDS:
            src = u16_block(rng, n, kind)
            for limit in (MAXSV, int(src.max()) if n else 0, 10):
                rr, rc, rm = ref.fse_count_u16(src, limit)
                r, c, m = hip.fse_count_u16(src, limit)
                assert r == rr, (n, kind, limit, r, rr)
                if not is_error(rr):
                    assert m == rm and (c[:limit + 1] == rc[:limit + 1]).all(), (n, kind, limit)
    r, _, _ = hip.fse_count_u16(u16_block(rng, 100, "p8"), 300)
    assert is_error(r)                                  # the device count array is the reference's: 287 entries


def test_compress_u16_matches_reference(hip, ref):
    rng = np.random.default_rng(2)
    checked = 0
    for n in (0, 1, 2, 3, 4, 5, 7, 8, 63, 64, 65, 1000, 4097, 16384, 40000):
        for kind in KINDS + ("rle",):
            src = u16_block(rng, n, kind)
            for tl in (0, 5, 9, 12, 13):
                for msv in (0, MAXSV):
                    rr, rout = ref.fse_compress_u16(src, msv, tl)
                    r, out = hip.fse_compress_u16(src, msv, tl)
                    assert r == rr, (n, kind, tl, msv, r, rr)
                    if not is_error(rr) and rr > 1:
                        assert (out[:rr] == rout[:rr]).all(), (n, kind, tl, msv)
                        checked += 1
    assert checked > 200


def test_compress_u16_argument_errors(hip, ref):
    rng = np.random.default_rng(3)
    src = u16_block(rng, 3000, "p8")
    for msv, tl in ((287, 12), (4000, 0), (0, 14), (MAXSV, 20), (200, 12), (1, 0)):   # limits too large; limits below the data
        rr, _ = ref.fse_compress_u16(src, msv, tl)
        r, _ = hip.fse_compress_u16(src, msv, tl)
        assert r == rr and is_error(r), (msv, tl, r, rr)


def test_compress_u16_small_destinations(hip, ref):
    rng = np.random.default_rng(4)
    for kind in ("p8", "p80", "flat"):
        src = u16_block(rng, 6000, kind)
        full, fout = ref.fse_compress_u16(src, 0, 0)
        assert full > 1 and not is_error(full)
        hdr = ref.fse_read_ncount(fout[:full], MAXSV)[0]
        assert not is_error(hdr) and hdr < full
        for cap in list(range(0, hdr + 12, 3)) + [full - 9, full - 8, full - 1, full, full + 7, full + 8, full + 9, full + 100]:
            r, out = hip.fse_compress_u16(src, 0, 0, cap=cap)
            if hdr <= cap <= hdr + 8:                    # undefined in the reference (it writes in front of dst): header size, no payload
                assert r == hdr or is_error(r), (kind, cap, r, hdr)   # (error: the careful header writer may still refuse this capacity)
                continue
            rr, rout = ref.fse_compress_u16(src, 0, 0, cap=cap)
            assert r == rr, (kind, cap, r, rr)
            if not is_error(rr) and rr == full:
                assert (out[:rr] == rout[:rr]).all(), (kind, cap)


def test_decompress_u16_matches_reference(hip, ref):
    rng = np.random.default_rng(5)
    for n in (2, 3, 9, 100, 4097, 16384, 40000):
        for kind in KINDS:
            src = u16_block(rng, n, kind)
            for tl in (0, 6, 13):
                cs, comp = ref.fse_compress_u16(src, 0, tl)
                if is_error(cs) or cs <= 1:
                    continue
                comp = comp[:cs]
                for cap in (n, n + 1, n + 17, n - 1, max(n // 2, 1)):
                    rr, rout = ref.fse_decompress_u16(comp, cap)
                    r, out = hip.fse_decompress_u16(comp, cap)
                    assert r == rr, (n, kind, tl, cap, r, rr)
                    if not is_error(rr):
                        assert (out[:rr] == rout[:rr]).all() and (rr != n or (out[:n] == src).all()), (n, kind, tl, cap)


def test_decompress_u16_damaged_streams(hip, ref):
    rng = np.random.default_rng(6)
    compared = 0
    for kind in KINDS:
        src = u16_block(rng, 5000, kind)
        cs, comp = ref.fse_compress_u16(src, 0, 0)
        if is_error(cs) or cs <= 1:
            continue
        comp = comp[:cs].copy()
        hdr = ref.fse_read_ncount(comp, MAXSV)[0]
        assert not is_error(hdr) and hdr < cs
        for trial in range(40):
            bad = comp.copy()
            mode = trial % 4
            if mode == 0:                                 # payload bytes flipped
                for _ in range(1 + trial // 8):
                    bad[int(rng.integers(hdr, cs))] ^= int(rng.integers(1, 256))
            elif mode == 1:                               # truncated behind the header
                bad = bad[:int(rng.integers(hdr + 1, cs))]
            elif mode == 2:                               # last byte (end mark) damaged
                bad[-1] = int(rng.integers(0, 256))
            else:                                         # garbage payload
                bad[hdr:] = rng.integers(0, 256, cs - hdr, dtype=np.uint8)
            for cap in (5000, 5100, 4000):
                rr, rout = ref.fse_decompress_u16(bad, cap)
                r, out = hip.fse_decompress_u16(bad, cap)
                assert r == rr, (kind, trial, cap, r, rr)
                if not is_error(rr):
                    assert (out[:rr] == rout[:rr]).all(), (kind, trial, cap)
                compared += 1
        # header damage: compared only where the reference is defined (some payload left behind whatever header it parses)
        for trial in range(40):
            bad = comp.copy()
            bad[int(rng.integers(0, hdr))] ^= int(rng.integers(1, 256))
            h2 = ref.fse_read_ncount(bad, MAXSV)[0]
            if not is_error(h2) and h2 >= bad.size:
                r, _ = hip.fse_decompress_u16(bad, 5000)
                assert is_error(r)                        # (the reference dereferences a null pointer here)
                continue
            rr, rout = ref.fse_decompress_u16(bad, 5000)
            r, out = hip.fse_decompress_u16(bad, 5000)
            assert r == rr, (kind, "header", trial, r, rr)
            compared += 1
    assert compared > 300
    for size in (0, 1):                                   # fseU16.c:317
        r, _ = hip.fse_decompress_u16(np.zeros(size, np.uint8), 10)
        assert r == ref.fse_decompress_u16(np.zeros(size, np.uint8), 10)[0] and is_error(r)


def _write_ncount(norm, max_sv, tl):
    """The NCount header (format: SURVEY A.2, written by lib/fse_compress.c:227-320) for any table log -- the reference's writer is
    the byte coder's object and refuses 13.  Checked against the reference's bytes at table log 12 by the test below."""
    out = bytearray()
    bit_stream, bit_count = tl - 5, 4
    remaining, threshold, nb = (1 << tl) + 1, 1 << tl, tl + 1
    sym, prev0, alphabet = 0, False, max_sv + 1

    def flush16():
        nonlocal bit_stream, bit_count
        out.append(bit_stream & 0xFF); out.append((bit_stream >> 8) & 0xFF)
        bit_stream >>= 16; bit_count -= 16
    while sym < alphabet and remaining > 1:
        if prev0:
            start = sym
            while sym < alphabet and norm[sym] == 0:
                sym += 1
            assert sym < alphabet
            while sym >= start + 24:
                start += 24
                bit_stream += 0xFFFF << bit_count
                bit_count += 16; flush16()
            while sym >= start + 3:
                start += 3
                bit_stream += 3 << bit_count; bit_count += 2
            bit_stream += (sym - start) << bit_count; bit_count += 2
            if bit_count > 16:
                flush16()
        count = int(norm[sym]); sym += 1
        mx = (2 * threshold - 1) - remaining
        remaining -= abs(count)
        count += 1
        if count >= threshold:
            count += mx
        bit_stream += count << bit_count
        bit_count += nb - (1 if count < mx else 0)
        prev0 = count == 1
        assert remaining >= 1
        while remaining < threshold:
            nb -= 1; threshold >>= 1
        if bit_count > 16:
            flush16()
    assert remaining == 1
    out.append(bit_stream & 0xFF); out.append((bit_stream >> 8) & 0xFF)
    return np.frombuffer(bytes(out[:len(out) - 2 + (bit_count + 7) // 8]), dtype=np.uint8).copy()


def test_decompress_u16_table_log_13(hip, ref):
    """Streams with a table log of 13: the reference's compressor never writes one (its normaliser and header writer are the byte
    coder's, limit 12) but FSE_decompressU16 takes them (FSE_buildDTableU16 is instantiated with the 16-bit limits).  Built here from
    the reference's own parts: counts normalised to 12 bits and doubled, header by the writer above, table and payload by
    FSE_buildCTableU16 / FSE_compressU16_usingCTable.  Device side: k_u16_dprep's second launch and the lane-per-block decoder."""
    rng = np.random.default_rng(8)
    done = 0
    for tl, nsym in ((12, 200), (9, 60), (12, 256), (6, 20)):                # the header writer above == the reference's where that one works
        w = rng.random(nsym) ** 3
        c = np.maximum(np.floor(w / w.sum() * (1 << tl)), 1).astype(np.int64)
        c[np.argmax(c)] += (1 << tl) - int(c.sum())
        c[int(rng.integers(1, nsym - 1))] = 0; c[np.argmax(c)] += (1 << tl) - int(c.sum())     # a zero inside the alphabet (run coding)
        assert c.min() >= 0 and c[-1] > 0
        r, h = ref.fse_write_ncount(600, c.astype(np.int16), nsym - 1, tl)
        assert not is_error(r) and (h[:r] == _write_ncount(c, nsym - 1, tl)).all(), (tl, nsym)
    for n, kind in ((5000, "p8"), (16384, "p8"), (3000, "flat"), (20000, "p80"), (4097, "p8")):
        src = u16_block(rng, n, kind)
        msv = int(src.max())
        cnt = np.bincount(src, minlength=msv + 1).astype(np.uint32)
        # the byte coder's normaliser takes at most 256 symbols: do it here (largest-remainder on 12 bits, every present symbol >= 1)
        p = cnt.astype(np.float64) * 4096 / n
        norm12 = np.maximum(np.floor(p), (cnt > 0)).astype(np.int64)
        norm12[np.argmax(norm12)] += 4096 - int(norm12.sum())
        assert norm12.min() >= 0 and int(norm12.sum()) == 4096 and (norm12[cnt > 0] > 0).all()
        # check the header writer on this very distribution at table log 12 against the reference's (alphabets it accepts)
        if msv <= 255:
            r12, h12 = ref.fse_write_ncount(600, norm12.astype(np.int16), msv, 12)
            assert not is_error(r12) and (h12[:r12] == _write_ncount(norm12, msv, 12)).all()
        norm13 = (2 * norm12).astype(np.int16)
        hdr = _write_ncount(norm13, msv, 13)
        h, m2, tl2, back = ref.fse_read_ncount(hdr, MAXSV)
        assert h == hdr.size and tl2 == 13 and m2 == msv and (back[:msv + 1] == norm13).all()
        e, ct = ref.fse_build_ctable_u16(norm13, msv, 13)
        assert not is_error(e)
        cs, payload = ref.fse_compress_u16_using_ctable(src, ct, 2 * n + 1024)
        assert not is_error(cs) and cs > 0
        comp = np.concatenate([hdr, payload[:cs]])
        for cap in (n, n + 9, n - 1):
            rr, rout = ref.fse_decompress_u16(comp, cap)
            r, out = hip.fse_decompress_u16(comp, cap)
            assert r == rr, (n, kind, cap, r, rr)
            if not is_error(rr):
                assert (out[:rr] == rout[:rr]).all(), (n, kind, cap)
                if cap >= n:
                    assert rr == n and (rout[:n] == src).all()
                done += 1
        bad = comp.copy(); bad[int(rng.integers(hdr.size, comp.size))] ^= 0x10
        rr, rout = ref.fse_decompress_u16(bad, n)
        r, out = hip.fse_decompress_u16(bad, n)
        assert r == rr, (n, kind, "damaged", r, rr)
    assert done >= 8


@pytest.mark.parametrize("width", [9000, 9003, 2501])        # row strides of 18000 (16-byte aligned rows), 18006 and 5002 bytes (rows on every 2-byte phase of a 16-byte line)
def test_u16_batch_ragged(hip, ref, width):
    rng = np.random.default_rng(7)
    nb = 97
    sizes = rng.integers(0, width + 1, nb)
    sizes[:6] = (0, 1, 2, width, width, 3)
    host = np.zeros((nb, width), np.uint16)
    for b in range(nb):
        host[b, :sizes[b]] = u16_block(rng, int(sizes[b]), (KINDS + ("rle",))[b % 7])
    dev = torch.from_numpy(host.view(np.int16)).cuda()
    dsz = torch.from_numpy(sizes.astype(np.int64)).cuda()
    cdst, cres = hip.fse_compress_u16_batch(dev, table_log=0, max_symbol_value=0, sizes=dsz)
    counts, maxsv, cntres = hip.fse_count_u16_batch(dev, sizes=dsz)
    torch.cuda.synchronize()
    cdst_h, cres_h = cdst.cpu().numpy(), cres.cpu().numpy()
    for b in range(nb):
        rr, rout = ref.fse_compress_u16(host[b, :sizes[b]], 0, 0, cap=cdst.shape[1])
        assert int(cres_h[b]) == s64(rr), (b, sizes[b], cres_h[b], rr)
        if not is_error(rr) and rr > 1:
            assert (cdst_h[b, :rr] == rout[:rr]).all(), b
        cr, cc, cm = ref.fse_count_u16(host[b, :sizes[b]], MAXSV)
        assert int(cntres.cpu()[b]) == s64(cr) and int(maxsv.cpu()[b]) == cm and (counts.cpu().numpy()[b].astype(np.uint32) == cc[:MAXSV + 1]).all(), b
    # decode what compressed (results > 1), exact capacity per block is the uniform width: the decoder stops at its own end
    ok = cres > 1
    idx = torch.nonzero(ok).flatten()
    out, dres = hip.fse_decompress_u16_batch(cdst[idx], cres[idx], width)
    torch.cuda.synchronize()
    out_h, dres_h = out.cpu().numpy().view(np.uint16), dres.cpu().numpy()
    for k, b in enumerate(idx.cpu().numpy()):
        rr, rout = ref.fse_decompress_u16(cdst_h[b, :cres_h[b]], width)
        assert int(dres_h[k]) == s64(rr), (b, dres_h[k], rr)
        if not is_error(rr):
            assert (out_h[k, :rr] == rout[:rr]).all(), b
        if sizes[b] == width:
            assert int(dres_h[k]) == width and (out_h[k] == host[b]).all(), b


def test_reference_u16_fuzzer_on_device(hip):
    """programs/fuzzerU16.c compiled unmodified with FSE_countU16 / FSE_compressU16 / FSE_decompressU16 bound to the device
    (oracle/Makefile target `fuzzers`): round trips, larger and smaller destinations, its count unit tests"""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "fuzzerU16-mi355x")
    if not os.path.exists(exe):
        pytest.skip("fuzzerU16-mi355x not built (needs the reference tree: make -C oracle fuzzers)")
    p = subprocess.run([exe, "-s1", "-i150"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "Error" not in out, out[-2000:]


def test_u16_decode_mixed_table_log_classes(hip, ref):
    """One FSE_decompressU16 batch whose blocks alternate between the decoder's table-log classes (csrc/fse_u16_decode.hip: 4 KiB slots for
    table logs up to 11, 8 KiB slots for 12): every workgroup of every launch finds blocks of the other
    classes in its range and must leave them alone.  Results and symbols against the compiled reference, block by block."""
    rng = np.random.default_rng(23)
    n, nb = 20000, 150
    logs = (12, 9, 11, 13, 12, 10, 12, 11, 0)
    host = np.stack([u16_block(rng, n, KINDS[b % len(KINDS)]) for b in range(nb)])
    dev = torch.from_numpy(host.view(np.int16)).cuda()
    rows, sizes = [None] * nb, [0] * nb
    for tl in sorted(set(logs)):
        idx = [b for b in range(nb) if logs[b % len(logs)] == tl]
        cd, cr = hip.fse_compress_u16_batch(dev[idx].contiguous(), table_log=tl)
        torch.cuda.synchronize()
        cdh, crh = cd.cpu().numpy(), cr.cpu().numpy()
        for k, b in enumerate(idx):
            rows[b], sizes[b] = cdh[k], int(crh[k])
    keep = [b for b in range(nb) if sizes[b] > 1]                       # (0 / 1: not compressible / one repeated symbol -- no stream to decode)
    assert len(keep) > 100
    seen = set()
    for b in keep:
        seen.add(int(rows[b][0]) & 15)                                  # the header's first four bits: table log - 5
    assert {4, 5, 6, 7} <= seen, seen                                   # table logs 9 ... 12 are all present (13 asked for: capped at 12 for 20000 symbols)
    cdst = torch.from_numpy(np.stack([rows[b] for b in keep])).cuda()
    cres = torch.tensor([sizes[b] for b in keep], dtype=torch.int64, device="cuda")
    out, dres = hip.fse_decompress_u16_batch(cdst, cres, n)
    torch.cuda.synchronize()
    out_h, dres_h = out.cpu().numpy().view(np.uint16), dres.cpu().numpy()
    for k, b in enumerate(keep):
        rr, rout = ref.fse_decompress_u16(rows[b][:sizes[b]], n)
        assert int(dres_h[k]) == s64(rr) == n, (b, dres_h[k], rr)
        assert (out_h[k, :n] == rout[:n]).all() and (out_h[k, :n] == host[b]).all(), b
