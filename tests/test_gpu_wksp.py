"""The `_wksp` entry points the reference's own callers go through (SURVEY 8(b) "what calls it"; lib/fse.h:315,335, lib/huf.h:95,164,289,
lib/hist.h:46,54) as drop-in names of libfsehip.so: same arguments into FSEHIP_<name> (device) and <name> of the compiled reference,
same return value, same bytes -- including what the reference makes of a workspace that is too small or misaligned."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SZ, VP, U = C.c_size_t, C.c_void_p, C.c_uint


def _p(a):
    return a.ctypes.data_as(VP)


def _both(hip, ref, name, *args):
    """call FSEHIP_<name> and the reference's <name> with the same arguments (fresh copies of the array arguments for each side)"""
    out = []
    for lib, fname in ((hip.lib, "FSEHIP_" + name), (ref.lib, name)):
        f = getattr(lib, fname)
        f.restype = SZ
        f.argtypes = None
        live = [a.copy() if isinstance(a, np.ndarray) else a for a in args]
        if os.environ.get("FSEHIP_TEST_TRACE"):
            print(fname, [a.shape if isinstance(a, np.ndarray) else a.value for a in live], flush=True)
        r = f(*[_p(a) if isinstance(a, np.ndarray) else a for a in live])
        out.append((int(r), [a for a in live if isinstance(a, np.ndarray)]))
    return out


def _blocks(checker):
    rng = np.random.default_rng(5)
    blocks = [checker.probagen_batch(P, 1, n, 7 + i)[0] for i, (P, n) in enumerate(((14, 32768), (80, 32768), (2, 32768), (14, 4097), (50, 1500), (20, 1499), (14, 300), (90, 12)))]
    blocks.append(np.full(5000, 7, np.uint8))                                   # one repeated byte -> 1
    blocks.append(rng.integers(0, 256, 20000, dtype=np.uint8))                  # noise -> 0
    blocks.append(np.zeros(1, np.uint8))
    blocks.append((rng.integers(0, 2, 3000, dtype=np.uint8) * 131).astype(np.uint8))
    return blocks


def test_hist_whole_buffer(hip, ref, checker):
    """HIST_count / HIST_countFast of ONE large input (lib/hist.c:163-180): from 256 KiB the device cuts it into 64 KiB pieces counted as a
    batch and folded (hist.hip k_hist_fold) -- sizes either side of the switch, ragged tails, a limit that is too small, a symbol that
    only the last byte holds."""
    rng = np.random.default_rng(11)
    ws = np.zeros(1024, np.uint32)
    for n, P in ((262143, 14), (262144, 80), (262145, 2), (1 << 20, 14), ((1 << 20) + 17, 50), (5 * 65536 + 65535, 20), (3000001, 80)):
        src = np.concatenate([checker.probagen_batch(P, 1, min(n - k, 1 << 20), 3 + k)[0] for k in range(0, n, 1 << 20)])
        assert src.size == n
        variants = [src]
        last = src.copy(); last[-1] = 255                   # the largest present symbol sits in the tail piece only
        variants.append(last)
        for v in variants:
            for limit in (255, 254, int(v.max()), int(v.max()) - 1, 30):
                for name, extra in (("HIST_count", ()), ("HIST_count_wksp", (ws, SZ(4096))), ("HIST_countFast", ())):
                    cnt = np.full(256 + 8, 0xABCDEF, np.uint32)
                    msv = np.array([limit], np.uint32)
                    (rg, ag), (rr, ar) = _both(hip, ref, name, cnt, msv, v, SZ(v.size), *extra)
                    assert rg == rr, (name, n, limit, rg, rr)
                    if rg < (1 << 64) - 200:
                        assert (ag[0] == ar[0]).all() and ag[1][0] == ar[1][0], (name, n, limit)
    noise = rng.integers(0, 256, 700000, dtype=np.uint8)
    cnt, msv = np.zeros(256, np.uint32), np.array([255], np.uint32)
    (rg, ag), (rr, ar) = _both(hip, ref, "HIST_count", cnt, msv, noise, SZ(noise.size))
    assert rg == rr and (ag[0] == ar[0]).all() and int(ag[0].sum()) == noise.size


def test_hist_wksp_and_fast(hip, ref, checker):
    ws = np.zeros(1024 + 1, np.uint32)
    for src in _blocks(checker):
        for limit in (255, 254, 200, 131, 52, 6):
            for name, extra in (("HIST_count_wksp", (ws, SZ(4096))), ("HIST_countFast", ()), ("HIST_countFast_wksp", (ws, SZ(4096))), ("HIST_count_simple", ())):
                if name != "HIST_count_wksp" and (src.size < 1500 or name == "HIST_count_simple") and int(src.max()) > limit:
                    continue                        # HIST_count_simple writes beyond count[] there (lib/hist.c:40): nothing to compare with
                cnt = np.zeros(256 + 8, np.uint32)
                msv = np.array([limit], np.uint32)
                (rg, ag), (rr, ar) = _both(hip, ref, name, cnt, msv, src, SZ(src.size), *extra)
                if name == "HIST_count_simple":          # returns `unsigned`: the upper half of the register is not part of the result
                    rg, rr = rg & 0xFFFFFFFF, rr & 0xFFFFFFFF
                assert rg == rr, (name, src.size, limit, rg, rr)
                if rg < (1 << 64) - 9:
                    assert (ag[0] == ar[0]).all() and ag[1][0] == ar[1][0], (name, src.size, limit)
    # the workspace checks of lib/hist.c:168-169, in their order
    src = _blocks(checker)[0]
    cnt, msv = np.zeros(256, np.uint32), np.array([255], np.uint32)
    raw = np.zeros(4200, np.uint8)
    for base_name in ("HIST_count_wksp", "HIST_countFast_wksp"):
        for off, size in ((1, 4096), (0, 4095), (2, 100), (0, 0)):
            view = raw[off:]
            res = []
            for lib, fname in ((hip.lib, "FSEHIP_" + base_name), (ref.lib, base_name)):
                f = getattr(lib, fname)
                f.restype = SZ
                res.append(int(f(_p(cnt), _p(msv), _p(src), SZ(src.size), C.c_void_p(view.ctypes.data), SZ(size))))
            assert res[0] == res[1] and res[0] > (1 << 64) - 9, (base_name, off, size, res)
    # below 1500 bytes HIST_countFast_wksp does not look at its workspace (lib/hist.c:145-146)
    small = _blocks(checker)[6]
    res = []
    for lib, fname in ((hip.lib, "FSEHIP_HIST_countFast_wksp"), (ref.lib, "HIST_countFast_wksp")):
        f = getattr(lib, fname)
        f.restype = SZ
        m2 = np.array([255], np.uint32)
        res.append(int(f(_p(cnt), _p(m2), _p(small), SZ(small.size), C.c_void_p(raw[1:].ctypes.data), SZ(0))))
    assert res[0] == res[1] and res[0] < (1 << 64) - 9, res


@pytest.mark.parametrize("table_log", [11, 12, 9, 13, 15])
def test_fse_compress_wksp(hip, ref, checker, table_log):
    """(Only calls the reference defines: FSE_compress_wksp carves its CTable out of the workspace at FSE_CTABLE_SIZE_U32(tableLog,
    maxSymbolValue) of the arguments AS PASSED (lib/fse_compress.c:640-642), so maxSymbolValue 0 ("default") or a tableLog that
    FSE_optimalTableLog has to raise (below highbit(maxSymbolValue) + 2) make FSE_buildCTable_wksp write into its own scratch -- the
    reference crashed on exactly that here.  FSE_compress2, which owns a full-size workspace, is where those arguments are compared.)"""
    ws = np.zeros(40000, np.uint32)         # FSE_WKSP_SIZE_U32(15, 255) words and more: the reference really uses it
    for src in _blocks(checker):
        for msv in (255, 52):
            cap = src.size + (src.size >> 7) + 600
            dst = np.zeros(cap + 8, np.uint8)
            (rg, ag), (rr, ar) = _both(hip, ref, "FSE_compress_wksp", dst, SZ(cap), src, SZ(src.size), U(msv), U(table_log), ws, SZ(4 * ws.size))
            assert rg == rr, (src.size, msv, table_log, rg, rr)
            if 1 < rg < (1 << 64) - 9:
                assert (ag[0][:rg] == ar[0][:rg]).all(), (src.size, msv, table_log)
    # lib/fse_compress.c:646: wkspSize (bytes) against FSE_WKSP_SIZE_U32 -> tableLog_tooLarge
    src = _blocks(checker)[0]
    dst = np.zeros(40000, np.uint8)
    need = 1 + (1 << (table_log - 1)) + 2 * 256 + ((1 << (table_log - 2)) if table_log > 12 else 1024)
    for size in (need - 1, need, 0):
        (rg, _), (rr, _) = _both(hip, ref, "FSE_compress_wksp", dst, SZ(40000), src, SZ(src.size), U(255), U(table_log), ws, SZ(size))
        assert rg == rr, (table_log, size, rg, rr)


def test_fse_decompress_wksp(hip, ref, checker):
    for src in _blocks(checker):
        for tl in (11, 12, 9):
            r, comp = checker.fse_compress2(src, 255, tl)
            if r <= 1 or r >= (1 << 63):
                continue
            comp = np.ascontiguousarray(comp[:r])
            for max_log in (12, 11, 10, 9, 6, 5):
                dt = np.zeros(1 + (1 << 12), np.uint32)
                out = np.zeros(src.size + 8, np.uint8)
                (rg, ag), (rr, ar) = _both(hip, ref, "FSE_decompress_wksp", out, SZ(src.size), comp, SZ(r), dt, U(max_log))
                assert rg == rr, (src.size, tl, max_log, rg, rr)
                if rg < (1 << 64) - 9:
                    assert (ag[0][:rg] == src[:rg]).all() and (ar[0][:rg] == src[:rg]).all()
                    h = int(ar[2][0]) & 0xFFFF
                    assert (ag[2][:1 + (1 << h)] == ar[2][:1 + (1 << h)]).all(), "the DTable left in the workspace differs"
            # truncated / damaged streams give the reference's verdicts
            for cut in (r - 1, r // 2, 3, 1):
                out = np.zeros(src.size + 8, np.uint8)
                dt = np.zeros(1 + (1 << 12), np.uint32)
                part = np.ascontiguousarray(comp[:cut])
                (rg, _), (rr, _) = _both(hip, ref, "FSE_decompress_wksp", out, SZ(src.size), part, SZ(cut), dt, U(12))
                assert rg == rr, (src.size, tl, cut, rg, rr)


@pytest.mark.parametrize("name", ["HUF_compress4X_wksp", "HUF_compress1X_wksp"])
def test_huf_compress_wksp(hip, ref, checker, name):
    ws = np.zeros((6 << 10) // 4 + 64 + 1, np.uint32)
    for src in _blocks(checker):
        for msv, tl in ((255, 11), (0, 0), (255, 12), (200, 8), (255, 13), (256, 11)):
            cap = src.size + (src.size >> 8) + 140
            dst = np.zeros(cap + 8, np.uint8)
            (rg, ag), (rr, ar) = _both(hip, ref, name, dst, SZ(cap), src, SZ(src.size), U(msv), U(tl), ws, SZ((6 << 10) + 256))
            assert rg == rr, (name, src.size, msv, tl, rg, rr)
            if 0 < rg < (1 << 64) - 9:
                assert (ag[0][:rg] == ar[0][:rg]).all(), (name, src.size, msv, tl)
    # lib/huf_compress.c:654-655: alignment first, then size
    src = _blocks(checker)[0]
    dst = np.zeros(40000, np.uint8)
    raw = np.zeros(8000, np.uint8)
    for off, size in ((1, 6400), (0, 6399), (2, 10), (0, 6400)):
        res = []
        for lib, fname in ((hip.lib, "FSEHIP_" + name), (ref.lib, name)):
            f = getattr(lib, fname)
            f.restype = SZ
            res.append(int(f(_p(dst), SZ(40000), _p(src), SZ(src.size), U(255), U(11), C.c_void_p(raw[off:].ctypes.data), SZ(size))))
        assert res[0] == res[1], (name, off, size, res)


def test_huf_decompress4x1_dctx_wksp(hip, ref, checker):
    ws = np.zeros(512, np.uint32)
    for src in _blocks(checker):
        for tl in (11, 12, 8):
            r, comp = checker.huf_compress2(src, 255, tl)
            if r <= 1 or r >= (1 << 63):
                continue
            comp = np.ascontiguousarray(comp[:r])
            for max_tl in (12, 11, 9, 6):                         # HUF_CREATE_STATIC_DTABLEX1(DTable, max_tl): descriptor = (max_tl - 1) * 0x01000001
                dctx = np.zeros(1 + (1 << 11), np.uint32)
                dctx[0] = (max_tl - 1) * 0x01000001
                out = np.zeros(src.size + 8, np.uint8)
                (rg, ag), (rr, ar) = _both(hip, ref, "HUF_decompress4X1_DCtx_wksp", dctx, out, SZ(src.size), comp, SZ(r), ws, SZ(2048))
                assert rg == rr, (src.size, tl, max_tl, rg, rr)
                if rg < (1 << 64) - 9:
                    assert (ag[1][:rg] == src[:rg]).all()
                    htl = (int(ar[0][0]) >> 16) & 0xFF
                    assert (ag[0][:1 + (1 << htl) // 2] == ar[0][:1 + (1 << htl) // 2]).all(), "the DTable left in dctx differs"
            for cut, wsz in ((r - 1, 2048), (r // 2, 2048), (2, 2048), (r, 319), (r, 320)):
                dctx = np.zeros(1 + (1 << 11), np.uint32)
                dctx[0] = 11 * 0x01000001
                out = np.zeros(src.size + 8, np.uint8)
                part = np.ascontiguousarray(comp[:cut])
                (rg, _), (rr, _) = _both(hip, ref, "HUF_decompress4X1_DCtx_wksp", dctx, out, SZ(src.size), part, SZ(cut), ws, SZ(wsz))
                assert rg == rr, (src.size, tl, cut, wsz, rg, rr)


def test_huf_single_symbol_family_under_the_reference_names(hip, ref, checker):
    """HUF_readDTableX1[_wksp], HUF_decompress4X1[_DCtx], HUF_decompress1X1[_DCtx[_wksp]] (lib/huf.h:141-143,161-163,209-211,299-304): the same
    arguments into the device call and the reference -- results, regenerated bytes and the table left in the caller's DTable / DCtx; whole,
    truncated and damaged blocks; table-log limits below the block's table; workspaces one byte short"""
    ws = np.zeros(512, np.uint32)
    rng = np.random.default_rng(9)
    checked = 0
    for src in _blocks(checker):
        for tl in (11, 8):
            for streams in (4, 1):
                comp = np.zeros(src.size + src.size // 2 + 600, np.uint8)
                (r, ag), (rr, ar) = _both(hip, ref, "HUF_compress2" if streams == 4 else "HUF_compress1X", comp, SZ(comp.size), src, SZ(src.size), U(255), U(tl))
                assert r == rr and (r >= (1 << 63) or (ag[0][:r] == ar[0][:r]).all()), (streams, src.size, tl, r, rr)
                comp = ar[0]
                if r <= 1 or r >= (1 << 63):
                    continue
                comp = np.ascontiguousarray(comp[:r])
                fam = "4X1" if streams == 4 else "1X1"
                variants = [(comp, r)] + [(np.ascontiguousarray(comp[:cut]), cut) for cut in (r - 1, r // 2, 3)]
                hit = comp.copy(); hit[int(rng.integers(0, r))] ^= 1 << int(rng.integers(0, 8))
                variants.append((hit, r))
                for part, n in variants:
                    out = np.zeros(src.size + 8, np.uint8)
                    (rg, ag), (rr, ar) = _both(hip, ref, "HUF_decompress" + fam, out, SZ(src.size), part, SZ(n))
                    assert rg == rr, (fam, src.size, tl, n, rg, rr)
                    if rg < (1 << 64) - 9:
                        assert (ag[0][:rg] == ar[0][:rg]).all(), (fam, src.size, tl, n)
                        checked += 1
                    for max_tl in (12, 9, 6):
                        dctx = np.zeros(1 + (1 << 11), np.uint32)
                        dctx[0] = (max_tl - 1) * 0x01000001
                        out = np.zeros(src.size + 8, np.uint8)
                        (rg, ag), (rr, ar) = _both(hip, ref, "HUF_decompress%s_DCtx" % fam, dctx, out, SZ(src.size), part, SZ(n))
                        assert rg == rr, (fam, "DCtx", src.size, tl, max_tl, n, rg, rr)
                        if rg < (1 << 64) - 9:
                            htl = (int(ar[0][0]) >> 16) & 0xFF
                            assert (ag[1][:rg] == ar[1][:rg]).all() and (ag[0][:1 + (1 << htl) // 2] == ar[0][:1 + (1 << htl) // 2]).all()
                    if streams == 1:
                        for wsz in (2048, 320, 319):
                            dctx = np.zeros(1 + (1 << 11), np.uint32)
                            dctx[0] = 11 * 0x01000001
                            out = np.zeros(src.size + 8, np.uint8)
                            (rg, _), (rr, _) = _both(hip, ref, "HUF_decompress1X1_DCtx_wksp", dctx, out, SZ(src.size), part, SZ(n), ws, SZ(wsz))
                            assert rg == rr, ("1X1_DCtx_wksp", src.size, tl, n, wsz, rg, rr)
                    # the table alone
                    for max_tl in (12, 10, 7):
                        dt = np.zeros(1 + (1 << 11), np.uint32)
                        dt[0] = (max_tl - 1) * 0x01000001
                        (rg, ag), (rr, ar) = _both(hip, ref, "HUF_readDTableX1", dt, part, SZ(n))
                        assert rg == rr, ("readDTableX1", src.size, tl, max_tl, n, rg, rr)
                        if rg < (1 << 64) - 9:
                            htl = (int(ar[0][0]) >> 16) & 0xFF
                            assert (ag[0][:1 + (1 << htl) // 2] == ar[0][:1 + (1 << htl) // 2]).all(), ("readDTableX1", src.size, tl, max_tl)
                    for wsz in (320, 319):
                        dt = np.zeros(1 + (1 << 11), np.uint32)
                        dt[0] = 11 * 0x01000001
                        (rg, _), (rr, _) = _both(hip, ref, "HUF_readDTableX1_wksp", dt, part, SZ(n), ws, SZ(wsz))
                        assert rg == rr, ("readDTableX1_wksp", wsz, rg, rr)
    assert checked > 20


def test_huf_build_and_write_ctable_under_the_reference_names(hip, ref, checker):
    """HUF_buildCTable[_wksp] on the caller's counters and HUF_writeCTable on the caller's table (lib/huf.h:204-218): the same arguments into the
    device call and the reference -- table logs, codes, header bytes; histograms of real blocks, flat and steep ones (trees deeper than the limit:
    the cut and its repair), a single symbol in use, limits 0 (default) .. 15, workspaces misaligned and one byte short, destinations too small"""
    rng = np.random.default_rng(21)
    hists = []
    for src in _blocks(checker):
        c = np.bincount(src, minlength=256).astype(np.uint32)
        hists.append((c, int(np.nonzero(c)[0].max())))
    fib = np.zeros(256, np.uint32); a, b = 1, 1
    for i in range(30):
        fib[3 * i] = a; a, b = b, a + b                         # steep: the unlimited tree is 29 levels deep
    hists.append((fib, 87))
    hists.append((rng.integers(1, 5, 256).astype(np.uint32), 255))                       # flat: every length 8
    hists.append((np.where(rng.random(256) < 0.1, rng.integers(1, 100000, 256), 0).astype(np.uint32), 255))
    one = np.zeros(256, np.uint32); one[77] = 1234
    hists.append((one, 77)); hists.append((one, 200))
    two = np.zeros(256, np.uint32); two[3] = 5; two[250] = 5
    hists.append((two, 250))
    wksp = np.zeros(1200, np.uint32)
    built = []
    for count, msv in hists:
        used = int((count[:msv + 1] != 0).sum())
        for limit in (0, 11, 12, 8, 5, 15, 3):
            lim = limit or 11
            if used == 0 or (lim <= 12 and used > (1 << lim)):
                continue                                          # undefined in the reference (refused on the device: below)
            tree = np.zeros(256, np.uint32)
            if limit > 12:
                # a limit above HUF_TABLELOG_MAX: the reference returns GENERIC when the tree is deeper than 12 (lib/huf_compress.c:385) -- but its
                # HUF_setMaxHeight has overrun rankLast[] by then when the tree is deeper than the limit by 13 or more ("stack smashing detected"
                # with the steep histogram), so the reference is only asked what a tree of at most 12 levels gives, at limit 12
                f = hip.lib.FSEHIP_HUF_buildCTable; f.restype = SZ
                g = tree.copy()
                rg = int(f(_p(g), _p(count), U(msv), U(limit)))
                (r12, a12), _ = _both(hip, ref, "HUF_buildCTable", tree, count, U(msv), U(12))
                deep = bool((a12[0][:msv + 1] >> 16).max() == 12) and used > 12       # (12 after a cut, or exactly 12 levels: then both answers are right)
                assert (rg == r12 and (g == a12[0]).all()) or (deep and rg == (1 << 64) - 1), ("HUF_buildCTable", msv, used, limit, rg, r12)
                continue
            (rg, ag), (rr, ar) = _both(hip, ref, "HUF_buildCTable", tree, count, U(msv), U(limit))
            assert rg == rr, ("HUF_buildCTable", msv, used, limit, rg, rr)
            if rg < (1 << 64) - 200:
                assert (ag[0][:msv + 1] == ar[0][:msv + 1]).all(), ("HUF_buildCTable", msv, used, limit, np.nonzero(ag[0][:msv + 1] != ar[0][:msv + 1])[0][:8])
                built.append((ar[0].copy(), msv, rr))
        tree = np.zeros(256, np.uint32)
        for off, size in ((0, 4352), (0, 4351), (1, 4400), (0, 4800)):
            res = []
            for lib, fname in ((hip.lib, "FSEHIP_HUF_buildCTable_wksp"), (ref.lib, "HUF_buildCTable_wksp")):
                f = getattr(lib, fname); f.restype = SZ
                raw = wksp.view(np.uint8)
                res.append(int(f(_p(tree.copy()), _p(count), U(msv), U(11), C.c_void_p(raw[off:].ctypes.data), SZ(size))))
            assert res[0] == res[1] or used == 0, ("HUF_buildCTable_wksp", off, size, res)
    assert len(built) > 40
    generic = (1 << 64) - 1
    tree = np.zeros(256, np.uint32)
    f = hip.lib.FSEHIP_HUF_buildCTable; f.restype = SZ
    assert int(f(_p(tree), _p(np.zeros(256, np.uint32)), U(255), U(11))) == generic                      # no symbol in use
    assert int(f(_p(tree), _p(np.ones(256, np.uint32)), U(255), U(7))) == generic                        # 256 symbols, codes of 7 bits
    assert int(f(_p(tree), _p(np.ones(256, np.uint32)), U(256), U(11))) == (1 << 64) - 6                  # maxSymbolValue_tooLarge (:350)
    for tree, msv, log in built:
        full = None
        for cap in (300, None, -1, -2, 5, 1):                 # (not 0: the reference hands maxDstSize - 1 to its weight coder, lib/huf_compress.c:133)
            if cap is None or (cap is not None and cap < 0):
                if full is None:
                    continue
                cap = full + (0 if cap is None else cap)
            dst = np.full(320, 0xA5, np.uint8)
            (rg, ag), (rr, ar) = _both(hip, ref, "HUF_writeCTable", dst, SZ(cap), tree, U(msv), U(log))
            assert rg == rr, ("HUF_writeCTable", msv, log, cap, rg, rr)
            if rg < (1 << 64) - 200:
                assert (ag[0][:rg] == ar[0][:rg]).all() and (ag[0][cap:] == 0xA5).all(), ("HUF_writeCTable", msv, log, cap)
                full = rg if full is None else full
    f = hip.lib.FSEHIP_HUF_writeCTable; f.restype = SZ
    tree, msv, log = built[0]
    dst = np.zeros(320, np.uint8)
    assert int(f(_p(dst), SZ(300), _p(tree), U(msv), U(max(log - 1, 0)))) == generic                     # a length above huffLog: beyond the reference's bitsToWeight[]
    assert int(f(_p(dst), SZ(300), _p(tree), U(256), U(log))) == (1 << 64) - 6
