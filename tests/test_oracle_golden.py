"""The CPU restatement (oracle/fse_oracle.c) against (a) SURVEY.md Appendix B known answers and
(b) the fixtures produced by the compiled reference (tests/golden/make_golden.py).  Runs anywhere."""
import numpy as np
import pytest

from oracle.oracle import is_error

# P, seed -> xxh64(src), maxSV, maxCount, FSE size, FSE xxh64, HUF size, HUF xxh64
APPENDIX_B = {
    (14, 1): ("e0b4b888653f17dc", 52, 4587, 17179, "d3c06f8967167512", 17287, "f21fa7cccaa19fba"),
    (14, 2): ("f21a4913f6e4d787", 52, 4567, 17139, "ab27a550b7b1643c", 17241, "6392e46853095739"),
    (14, 3): ("3be33565918ef7a5", 52, 4629, 17164, "d01f026af235da15", 17276, "7c13c8e4eeae1e91"),
    (80, 1): ("05ea78218d920edc", 6, 26250, 3709, "a4c0b98b2066e40e", 5133, "0780d86255c9fccc"),
    (80, 2): ("3b06a30c09612577", 6, 26264, 3685, "58e64e7e0a2ad8b7", 5123, "670b7813d8dbc03a"),
    (80, 3): ("9bf8071aa2e26798", 6, 26204, 3717, "82024b8e7fe9150e", 5137, "8f5e2da1ee4cd879"),
    (2, 1): ("bb911e4769efaaf8", 255, 656, 29015, "31e6a4473b89eae1", 28981, "c911909969d04df4"),
    (2, 2): ("ddb9ec0bd3f76cf2", 255, 675, 28996, "cbff575b6c218da5", 28969, "69b24dffeb0b0015"),
    (2, 3): ("34a194ca9e6ac46f", 255, 654, 29003, "dd8592541084be06", 28963, "4d84693753e11fc8"),
}


@pytest.fixture(scope="module")
def oracle(restatement):
    """here the restatement itself is under test (conftest.py: every other module's `oracle` is the compiled reference where present)"""
    return restatement


def hx(v):
    return "%016x" % v


def test_appendix_b_known_answers(oracle):
    for (P, seed), (h_src, msv, mc, fs, fh, hs, hh) in APPENDIX_B.items():
        blk = oracle.probagen_batch(P, 1, 32768, seed)[0]
        assert hx(oracle.xxh64(blk)) == h_src
        mx, m, cnt = oracle.hist_count(blk)
        assert (mx, m) == (mc, msv) and int(cnt.sum()) == 32768
        cs, out = oracle.fse_compress2(blk)
        assert cs == fs and hx(oracle.xxh64(out[:cs])) == fh
        ds, dec = oracle.fse_decompress(out[:cs], 32768)
        assert ds == 32768 and (dec == blk).all()
        cs, out = oracle.huf_compress2(blk)
        assert cs == hs and hx(oracle.xxh64(out[:cs])) == hh
        ds, dec = oracle.huf_decompress(out[:cs], 32768)
        assert ds == 32768 and (dec == blk).all()


def test_golden_blocks(oracle, golden):
    meta = golden["meta"]
    for i, row in enumerate(meta):
        P, seed, n, h_src, mx, msv, fc, fh, fc12, fh12, hc, hh = [int(v) for v in row]
        blk = oracle.probagen_batch(P, 1, n, seed)[0]
        assert oracle.xxh64(blk) == h_src
        a, b, cnt = oracle.hist_count(blk)
        assert (a, b) == (mx, msv) and (cnt == golden["count_%d" % i]).all()
        cs, out = oracle.fse_compress2(blk, 255, 11)
        assert cs == fc and oracle.xxh64(out[:cs]) == fh
        cs12, out12 = oracle.fse_compress2(blk, 255, 12)
        assert cs12 == fc12 and oracle.xxh64(out12[:cs12]) == fh12
        hs, hout = oracle.huf_compress2(blk, 255, 11)
        assert hs == hc and oracle.xxh64(hout[:hs]) == hh
        if "fse_%d" % i in golden:
            assert (out[:cs] == golden["fse_%d" % i]).all()
            assert (hout[:hs] == golden["huf_%d" % i]).all()
            if fc > 1:      # 0 = not compressible, 1 = RLE (lib/fse.h:62-65)
                ds, dec = oracle.fse_decompress(golden["fse_%d" % i], n)
                assert ds == n and (dec == blk).all()
            if hc > 1:
                ds, dec = oracle.huf_decompress(golden["huf_%d" % i], n)
                assert ds == n and (dec == blk).all()


def test_golden_tables(oracle, golden):
    meta = golden["meta"]
    seen = 0
    for i, row in enumerate(meta):
        if "tl_%d" % i not in golden:
            continue
        seen += 1
        P, seed, n = int(row[0]), int(row[1]), int(row[2])
        tl, msv, hs, mb, whs, cs = [int(v) for v in golden["tl_%d" % i]]
        blk = oracle.probagen_batch(P, 1, n, seed)[0]
        cnt = golden["count_%d" % i]
        assert oracle.fse_optimal_tablelog(11, n, msv, 2) == tl
        r, norm = oracle.fse_normalize_count(tl, cnt, n, msv)
        assert r == tl and (norm == golden["norm_%d" % i]).all()
        r, hdr = oracle.fse_write_ncount(512, norm, msv, tl)
        assert r == hs and (hdr[:hs] == golden["ncount_%d" % i]).all()
        r, m2, tl2, norm2 = oracle.fse_read_ncount(hdr[:hs])
        assert (r, m2, tl2) == (hs, msv, tl) and (norm2 == norm).all()
        _, ct = oracle.fse_build_ctable(norm, msv, tl)
        tt = 1 + (1 << (tl - 1))
        for s in range(msv + 1):
            if norm[s] == 0:
                ct[tt + 2 * s] = 0
        assert (ct == golden["ctable_%d" % i]).all()
        _, dt = oracle.fse_build_dtable(norm, msv, tl)
        assert (dt == golden["dtable_%d" % i]).all()
        r, payload = oracle.fse_compress_using_ctable(blk, ct)
        assert r == cs
        r, dec = oracle.fse_decompress_using_dtable(payload[:cs], dt, n)
        assert r == n and (dec == blk).all()
        hl = oracle.fse_optimal_tablelog(11, n, msv, 1)
        r, celt = oracle.huf_build_ctable(cnt, msv, hl)
        assert r == mb and ((celt[:msv + 1] & 0xFFFFFF) == golden["celt_%d" % i]).all()
        r, whdr = oracle.huf_write_ctable(256, celt, msv, mb)
        assert r == whs and (whdr[:whs] == golden["hufhdr_%d" % i]).all()
        r, hdt = oracle.huf_read_dtable_x1(whdr[:whs], 11)
        assert r == whs and (hdt[:1 + (1 << mb)] == golden["hufdt_%d" % i]).all()
    assert seen == 3


def test_error_convention(oracle):
    # lib/error_private.h:77-79 : (size_t)-code, error iff > (size_t)-9
    assert not is_error(0) and not is_error(32768) and is_error((1 << 64) - 1) and is_error((1 << 64) - 8)
    assert not is_error((1 << 64) - 9)
    blk = np.full(100, 255, dtype=np.uint8)
    r, _, _ = oracle.hist_count(blk, 254)            # lib/hist.c:128
    assert r == (1 << 64) - 7
