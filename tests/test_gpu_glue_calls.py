"""The reference's "advanced" flow under its own call names (lib/fse.h:107-163, :218-241), on the device: FSE_optimalTableLog,
FSE_normalizeCount, FSE_NCountWriteBound, FSE_writeNCount, FSE_readNCount, FSE_buildCTable(_wksp), FSE_buildDTable as single calls on
host pointers (include/fsehip.h; the names libfse_dropin.so exports at link level) and FSE_buildCTable / FSE_buildDTable on the caller's
counters as batch calls -- results, counters, header bytes and tables word for word against the COMPILED REFERENCE (`ref`), the
reference's own unit vectors (programs/fuzzer.c:325-417) included, then the whole flow end to end through the device's hot loops."""
import numpy as np
import pytest
import torch

from oracle.oracle import Ref, fse_ctable_u32, fse_dtable_u32, is_error
from test_gpu_fse import s64
from test_gpu_glue_vectors import REFERENCE_VECTORS

pytestmark = pytest.mark.gpu
GENERIC, TL_TOO_LARGE, MSV_TOO_LARGE = (1 << 64) - 1, (1 << 64) - 5, (1 << 64) - 6


@pytest.fixture(scope="module")
def ref():
    if not Ref.available():
        pytest.skip("oracle/_ref/libfse_ref.so not built (needs the reference tree: make -C oracle ref)")
    return Ref()


def histograms(rng, n):
    """count[256], total, maxSymbolValue of blocks of several shapes (the fuzzer's generators in spirit: programs/fuzzer.c:107-134)"""
    out = []
    for i in range(n):
        kind = i % 6
        size = int(rng.choice([40, 300, 2000, 32768, 100000]))
        if kind == 0:
            src = rng.integers(0, int(rng.integers(2, 257)), size)
        elif kind == 1:
            src = np.minimum(rng.geometric(float(rng.choice([0.02, 0.14, 0.5, 0.8])), size) - 1, 255)
        elif kind == 2:
            src = rng.integers(0, 4, size) * 60 + 3
        elif kind == 3:                                      # one dominant symbol and stragglers: counters of -1
            src = np.full(size, 9)
            idx = rng.integers(0, size, max(size // 200, 1))
            src[idx] = rng.integers(0, 256, idx.size)
        elif kind == 4:
            src = (rng.normal(128, float(rng.choice([2, 10, 40])), size)).clip(0, 255).astype(np.int64)
        else:
            src = np.concatenate([np.arange(256), rng.integers(0, 7, size)])
        c = np.bincount(src.astype(np.int64), minlength=256).astype(np.uint32)
        out.append((c, int(c.sum()), int(np.nonzero(c)[0].max())))
    return out


def test_optimal_tablelog_and_ncount_write_bound(hip, ref):
    for max_tl in (0, 5, 9, 11, 12, 14):
        for size in (2, 3, 17, 300, 1000, 4096, 32768, 1 << 20):
            for msv in (1, 2, 15, 16, 100, 255):
                assert hip.fse_optimal_tablelog(max_tl, size, msv) == ref.fse_optimal_tablelog(max_tl, size, msv, 2), (max_tl, size, msv)
    for msv in (0, 1, 17, 128, 255):
        for tl in (5, 9, 12):
            assert hip.fse_ncount_write_bound(msv, tl) == (512 if msv == 0 else (((msv + 1) * tl) >> 3) + 3)      # lib/fse_compress.c:186-190


def test_normalize_write_read_single_calls(hip, ref):
    rng = np.random.default_rng(11)
    cases = [(c, t, m) for c, t, m in histograms(rng, 36) if t > 1]
    checked = 0
    for count, total, msv in cases:
        for tl_req in (0, 5, 8, 11, 12, 13):
            tl = tl_req if tl_req in (0, 13) else ref.fse_optimal_tablelog(tl_req, total, msv, 2)
            rr, rnorm = ref.fse_normalize_count(tl, count, total, msv)
            r, norm = hip.fse_normalize_count(tl, count, total, msv)
            assert r == rr, (total, msv, tl, r, rr)
            if is_error(rr):
                continue
            assert (norm[:msv + 1] == rnorm[:msv + 1]).all(), (total, msv, tl)
            used = rr
            full, rhdr = ref.fse_write_ncount(512, rnorm, msv, used)
            for cap in (512, full + 1, full, full - 1, full - 2, 3, 1, 0):
                wr, whdr = ref.fse_write_ncount(cap, rnorm, msv, used)
                w, hdr = hip.fse_write_ncount(cap, norm, msv, used)
                assert w == wr, (total, msv, used, cap, w, wr)
                if not is_error(wr):
                    assert (hdr[:wr] == whdr[:wr]).all(), (total, msv, used, cap)
            for limit, cut in ((255, 0), (msv, 0), (max(msv - 1, 0), 0), (255, 1), (255, 2), (255, full - 1), (msv, full - 3)):
                h = rhdr[:max(full - cut, 0)]
                if h.size == 0:
                    continue
                er, emsv, etl, enorm = ref.fse_read_ncount(h, limit)
                gr, gmsv, gtl, gnorm = hip.fse_read_ncount(h, limit)
                assert gr == er, (total, msv, used, limit, cut, gr, er)
                if not is_error(er):
                    assert (gmsv, gtl) == (emsv, etl) and (gnorm[:limit + 1] == enorm[:limit + 1]).all(), (total, msv, used, limit, cut)
                    checked += 1
    assert checked > 150
    # the reference's own unit vectors, as written (programs/fuzzer.c:325-364), and its argument checks (:330, lib/fse_compress.c:434-438)
    for tl, count, total, msv in REFERENCE_VECTORS:
        rr, rnorm = ref.fse_normalize_count(tl, count, total, msv)
        r, norm = hip.fse_normalize_count(tl, count, total, msv)
        assert r == rr and not is_error(r) and (norm[:msv + 1] == rnorm[:msv + 1]).all(), (tl, total, msv)
    count, total, msv = cases[0]
    assert is_error(hip.fse_normalize_count(8, np.ones(257, np.uint32), 257, 256)[0])             # maxSymbolValue 256 ("max >= 1 << tableLog")
    assert hip.fse_normalize_count(4, count, total, msv)[0] == GENERIC and hip.fse_normalize_count(13, count, total, msv)[0] == TL_TOO_LARGE
    norm = ref.fse_normalize_count(11, count, total, msv)[1]
    assert hip.fse_write_ncount(512, norm, msv, 13)[0] == TL_TOO_LARGE and hip.fse_write_ncount(512, norm, msv, 4)[0] == GENERIC   # lib/fse_compress.c:281-282
    bad = norm.copy(); bad[0] += 1                                                                   # not a distribution: "remaining != 1", :276
    assert hip.fse_write_ncount(512, bad, msv, 11)[0] == ref.fse_write_ncount(512, bad, msv, 11)[0] == GENERIC


def small_norms(rng, tl, n):
    """n valid counter sets for a table of 1 << tl cells (any tl >= 1): random compositions with some -1 entries"""
    out = []
    ts = 1 << tl
    for i in range(n):
        k = int(rng.integers(1, min(ts, 40) + 1))            # symbols in use
        cuts = np.sort(rng.choice(np.arange(1, ts), k - 1, replace=False)) if k > 1 else np.array([], np.int64)
        parts = np.diff(np.concatenate([[0], cuts, [ts]])).astype(np.int64)
        msv = int(rng.integers(k - 1, 256))
        where = np.sort(rng.choice(msv + 1, k, replace=False))
        where[-1] = msv if i % 2 else where[-1]
        where = np.unique(where)
        if where.size < k:                                   # (the forced last symbol collided: drop one part into its neighbour)
            parts = np.concatenate([parts[:where.size - 1], [parts[where.size - 1:].sum()]])
        norm = np.zeros(256, np.int16)
        norm[where] = parts
        ones = np.nonzero(norm == 1)[0]
        if ones.size and i % 3 == 0:
            norm[ones[rng.integers(0, ones.size, max(ones.size // 2, 1))]] = -1
        out.append((norm, int(where.max())))
    return out


@pytest.mark.parametrize("tl", [2, 4, 5, 6, 9, 11, 12])
def test_build_ctable_and_dtable_on_caller_counters(hip, ref, tl):
    rng = np.random.default_rng(100 + tl)
    sets = small_norms(rng, tl, 24)
    if tl >= 5:                                              # ... and what FSE_normalizeCount really produces
        for count, total, msv in histograms(rng, 18):
            if total > 1:
                r, norm = ref.fse_normalize_count(tl, count, total, msv)
                if not is_error(r):
                    sets.append((norm, msv))
    norms = torch.from_numpy(np.stack([s[0] for s in sets])).cuda()
    msvs = torch.tensor([s[1] for s in sets], dtype=torch.int32, device="cuda")
    ct, cres = hip.fse_build_ctable_from_norm_batch(norms, msvs, tl)
    dt, dres = hip.fse_build_dtable_from_norm_batch(norms, msvs, tl)
    ct_h, dt_h = ct.cpu().numpy().view(np.uint32), dt.cpu().numpy().view(np.uint32)
    assert (cres == 0).all() and (dres == 0).all(), (cres.tolist(), dres.tolist())
    for i, (norm, msv) in enumerate(sets):
        rc, ect = ref.fse_build_ctable(norm, msv, tl)
        rd, edt = ref.fse_build_dtable(norm, msv, tl)
        assert rc == 0 and rd == 0
        wc, wd = fse_ctable_u32(tl, msv), fse_dtable_u32(tl)
        assert (ct_h[i][:wc] == ect[:wc]).all(), (tl, i, "ctable batch", np.nonzero(ct_h[i][:wc] != ect[:wc])[0][:8])
        assert (dt_h[i][:wd] == edt[:wd]).all(), (tl, i, "dtable batch", np.nonzero(dt_h[i][:wd] != edt[:wd])[0][:8])
        if i < 10:                                           # the single calls on host pointers
            r, g = hip.fse_build_ctable(norm, msv, tl)
            assert r == 0 and (g[:wc] == ect[:wc]).all(), (tl, i, "ctable")
            r, g = hip.fse_build_dtable(norm, msv, tl)
            assert r == 0 and (g[:wd] == edt[:wd]).all(), (tl, i, "dtable")
    norm, msv = sets[0]
    r, g = hip.fse_build_ctable(norm, msv, tl, wksp_bytes=1 << tl)                    # lib/fse_compress.c:86: tableSize bytes suffice ...
    assert r == 0 and (g[:fse_ctable_u32(tl, msv)] == ref.fse_build_ctable(norm, msv, tl)[1][:fse_ctable_u32(tl, msv)]).all()
    assert hip.fse_build_ctable(norm, msv, tl, wksp_bytes=(1 << tl) - 1)[0] == TL_TOO_LARGE   # ... one less does not


def test_build_tables_refuse_what_the_reference_leaves_undefined(hip, ref):
    rng = np.random.default_rng(5)
    norm, msv = small_norms(rng, 9, 1)[0]
    for builder in (hip.fse_build_ctable, hip.fse_build_dtable):
        assert builder(norm, msv, 13)[0] == TL_TOO_LARGE                             # lib/fse_decompress.c:84; the CTable builder's workspace, lib/fse_compress.c:86,172-176
        assert builder(norm, 256, 9)[0] == MSV_TOO_LARGE                             # lib/fse_decompress.c:83
        assert builder(norm, msv, 0)[0] == GENERIC
        for tl in (1, 3):                                                            # FSE_TABLESTEP(2) = 4, FSE_TABLESTEP(8) = 8: the reference never leaves cell 0
            assert builder(small_norms(rng, tl, 1)[0][0], 255, tl)[0] == GENERIC
        for delta in (1, -1):
            bad = norm.copy()
            bad[np.nonzero(bad > 1)[0][0]] += delta
            assert builder(bad, msv, 9)[0] == GENERIC                                # not 1 << tableLog cells
        bad = norm.copy(); bad[np.nonzero(bad == 0)[0][0] if (bad[:msv + 1] == 0).any() else 0] = -2
        assert builder(bad, msv, 9)[0] == GENERIC
    assert ref.fse_build_dtable(np.where(norm > 1, norm - 1, norm), msv, 9)[0] == GENERIC         # (the reference's own verdict where it has one, :107)
    # a batch in which only some rows are valid: the others report, the valid ones are built
    sets = small_norms(rng, 7, 6)
    rows = np.stack([s[0] for s in sets])
    rows[1, 0] += 5; rows[4] = 0
    norms = torch.from_numpy(rows).cuda()
    msvs = torch.tensor([s[1] for s in sets], dtype=torch.int32, device="cuda")
    ct, cres = hip.fse_build_ctable_from_norm_batch(norms, msvs, 7)
    dt, dres = hip.fse_build_dtable_from_norm_batch(norms, msvs, 7)
    for res in (cres.cpu().numpy(), dres.cpu().numpy()):
        assert [int(v) for v in res] == [0, s64(GENERIC), 0, 0, s64(GENERIC), 0]
    for i in (0, 2, 3, 5):
        assert (ct[i].cpu().numpy().view(np.uint32)[:fse_ctable_u32(7, sets[i][1])] == ref.fse_build_ctable(sets[i][0], sets[i][1], 7)[1][:fse_ctable_u32(7, sets[i][1])]).all()
        assert (dt[i].cpu().numpy().view(np.uint32)[:129] == ref.fse_build_dtable(sets[i][0], sets[i][1], 7)[1][:129]).all()
    cres13 = hip.fse_build_ctable_from_norm_batch(norms, msvs, 13)[1]
    dres13 = hip.fse_build_dtable_from_norm_batch(norms, msvs, 13)[1]
    assert (cres13 == s64(TL_TOO_LARGE)).all() and (dres13 == s64(TL_TOO_LARGE)).all()


def test_advanced_flow_end_to_end_on_the_device(hip, ref):
    """count -> table log -> normalise -> header -> CTable -> FSE_compress_usingCTable | FSE_readNCount -> FSE_buildDTable ->
    FSE_decompress_usingDTable, every step a device call under the reference's name and signature, every intermediate compared with the
    reference run on the same inputs (what programs/fullbench.c's FSE cases and a caller with its own block format do)"""
    rng = np.random.default_rng(77)
    for size, p, tl_req in ((32768, 0.14, 0), (32768, 0.8, 12), (5000, 0.02, 9), (700, 0.3, 0), (65536, 0.5, 11)):
        src = np.minimum(rng.geometric(p, size) - 1, 255).astype(np.uint8)
        mx, msv, count = hip.hist_count(src, 255)
        assert (mx, msv) == ref.hist_count(src, 255)[:2]
        tl = hip.fse_optimal_tablelog(tl_req, size, msv)
        r, norm = hip.fse_normalize_count(tl, count, size, msv)
        assert r == tl == ref.fse_normalize_count(tl, count, size, msv)[0]
        bound = hip.fse_ncount_write_bound(msv, tl)
        h, hdr = hip.fse_write_ncount(bound, norm, msv, tl)
        rh, rhdr = ref.fse_write_ncount(bound, norm, msv, tl)
        assert h == rh and (hdr[:h] == rhdr[:h]).all()
        r, ct = hip.fse_build_ctable(norm, msv, tl)
        assert r == 0
        c, comp = hip.fse_compress_using_ctable(src, ct)
        rc, rcomp = ref.fse_compress_using_ctable(src, ref.fse_build_ctable(norm, msv, tl)[1])
        assert c == rc and c > 1 and (comp[:c] == rcomp[:c]).all(), (size, p, tl)
        frame = np.concatenate([hdr[:h], comp[:c]])                                   # = what FSE_compress2 writes
        whole, wcomp = ref.fse_compress2(src, msv, tl_req or 11)
        assert whole == frame.size and (wcomp[:whole] == frame).all(), (size, p, tl_req)
        g, gmsv, gtl, gnorm = hip.fse_read_ncount(frame, 255)
        assert (g, gmsv, gtl) == (h, msv, tl) and (gnorm[:msv + 1] == norm[:msv + 1]).all()
        r, dt = hip.fse_build_dtable(gnorm, gmsv, gtl)
        assert r == 0 and (dt == ref.fse_build_dtable(gnorm, gmsv, gtl)[1][:dt.size]).all()
        d, out = hip.fse_decompress_using_dtable(frame[g:], dt, size)
        assert d == size and (out[:size] == src).all(), (size, p, tl)
