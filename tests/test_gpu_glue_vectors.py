"""The reference's own unit vectors for the table glue (programs/fuzzer.c:325-364 FSE_normalizeCount, :367-417 FSE_writeNCount /
FSE_readNCount), put to the DEVICE routines exactly as written -- FSEHIP_FSE_normalizeCount_batch / _writeNCount_batch /
_readNCount_batch run the same wave code as k_fse_cprep / k_fse_dparse (wave_glue.h:106-222, ncount_reader.h) -- and then, where the
one-shot path can reach them, through FSEHIP_FSE_buildCTable_batch / FSEHIP_FSE_compress_batch on blocks synthesised to have exactly
those histograms.  Everything is compared with the compiled reference (`checker`)."""
import numpy as np
import pytest
import torch

from oracle.oracle import fse_ctable_u32, is_error
from test_gpu_fse import s64

pytestmark = pytest.mark.gpu


def _vec(fill, head, upto):
    c = np.zeros(256, np.uint32)
    c[:upto + 1] = fill
    c[:len(head)] = head
    return c


# programs/fuzzer.c:336-363: (tableLog, count[], total, maxSymbolValue)
RANK_OVERFLOW_A = (10, _vec(6, (940, 910, 470, 190, 90), 255), 940 + 910 + 470 + 190 + 90 + 251 * 6, 255)
RANK_OVERFLOW_B = (10, _vec(4, (300, 300, 300, 300, 50), 80), 4 * 300 + 50 + 76 * 4, 80)
M2_DIV_BY_ZERO = (5, np.concatenate([np.zeros(22, np.uint32), np.ones(22, np.uint32), np.zeros(212, np.uint32)]), 22, 43)
REFERENCE_VECTORS = (RANK_OVERFLOW_A, RANK_OVERFLOW_B, M2_DIV_BY_ZERO)


def takes_m2(tl, count, total, msv):
    """whether FSE_normalizeCount leaves its main loop for FSE_normalizeM2 (lib/fse_compress.c:446-475): -stillToDistribute >= norm[largest] >> 1"""
    rtb = (0, 473195, 504333, 520860, 550000, 700000, 750000, 830000)
    scale, step, vstep, low = 62 - tl, (1 << 62) // total, 1 << (62 - tl - 20), total >> tl
    still, largest, largest_p = 1 << tl, 0, 0
    for s in range(msv + 1):
        c = int(count[s])
        if c == 0:
            continue
        if c <= low:
            still -= 1
            continue
        p = ((c * step) >> scale) & 0xFFFF
        if p < 8 and c * step - (p << scale) > vstep * rtb[p]:
            p += 1
        if p > largest_p:
            largest_p, largest = p, s
        still -= p
    return -still >= (largest_p >> 1)


def _normalize_on_device(hip, vecs):
    """vecs grouped by table log -> list of (result, norm[256]) in the order given"""
    out = [None] * len(vecs)
    for tl in sorted(set(v[0] for v in vecs)):
        sel = [i for i, v in enumerate(vecs) if v[0] == tl]
        counts = torch.from_numpy(np.stack([vecs[i][1] for i in sel]).astype(np.int32)).cuda()
        totals = torch.tensor([vecs[i][2] for i in sel], dtype=torch.int64, device="cuda")
        msvs = torch.tensor([vecs[i][3] for i in sel], dtype=torch.int32, device="cuda")
        norms, res = hip.fse_normalize_count_batch(counts, totals, msvs, tl)
        nh, rh = norms.cpu().numpy(), res.cpu().numpy()
        for k, i in enumerate(sel):
            out[i] = (int(rh[k]), nh[k])
    return out


def _compare_normalize(hip, checker, vecs):
    got = _normalize_on_device(hip, vecs)
    for (tl, c, tot, msv), (r, norm) in zip(vecs, got):
        er, en = checker.fse_normalize_count(tl, c, tot, msv)
        assert r == s64(er), (tl, tot, msv, r, er)
        if not is_error(er):
            assert er == (tl or 11)
            assert (norm[:msv + 1] == en[:msv + 1]).all(), (tl, tot, msv, np.nonzero(norm[:msv + 1] != en[:msv + 1])[0][:8])
            assert int(np.abs(norm[:msv + 1].astype(np.int32)).sum()) == 1 << er


def test_normalize_count_reference_unit_vectors(hip, checker):
    """programs/fuzzer.c:336-363 exactly: all three must work ("should have worked"); the first and the third leave the main loop for
    FSE_normalizeM2 (the second is settled by the main loop's correction of the largest counter); the device's counters are the reference's"""
    assert [takes_m2(*v) for v in REFERENCE_VECTORS] == [True, False, True]
    for tl, c, tot, msv in REFERENCE_VECTORS:
        assert not is_error(checker.fse_normalize_count(tl, c, tot, msv)[0])
    _compare_normalize(hip, checker, list(REFERENCE_VECTORS))
    # :329-332: 16 KB of noise at tableLog 10 works, at tableLog 8 with maxSymbolValue 256 -> an alphabet of 256 does not fit: error
    rng = np.random.default_rng(5)
    c = np.bincount(rng.integers(0, 256, 16384), minlength=256).astype(np.uint32)
    _compare_normalize(hip, checker, [(10, c, 16384, 255), (8, c, 16384, 255), (0, c, 16384, 255), (4, c, 16384, 255), (13, c, 16384, 255), (12, c, 16384, 255)])


def test_normalize_count_flat_and_near_flat_sweep_forces_m2(hip, checker):
    """flat / near-flat histograms (the shape of the three vectors above, swept): most of them take FSE_normalizeM2 -- every branch of
    wg_normalize_fallback (second widening of "about one point", nothing pending, round robin, proportional share) against the reference"""
    rng = np.random.default_rng(2024)
    vecs = []
    for _ in range(1500):
        tl = int(rng.choice([5, 6, 7, 8, 9, 10, 11, 12]))
        k = int(rng.integers(2, min(256, 1 << tl) + 1))                       # symbols in use
        shape = int(rng.integers(0, 6))
        c = np.zeros(256, np.uint32)
        base = int(rng.choice([1, 2, 3, 4, 6, 10, 40]))
        if shape == 0:                                                         # perfectly flat
            c[:k] = base
        elif shape == 1:                                                       # flat with a few heads
            c[:k] = base
            h = int(rng.integers(1, min(k, 6) + 1))
            c[:h] = rng.integers(base, base * 200 + 2, h)
        elif shape == 2:                                                       # flat, offset alphabet with holes in front (the :358-362 vector's shape)
            lo = int(rng.integers(0, 256 - k + 1))
            c[lo:lo + k] = base
        elif shape == 3:                                                       # two plateaus
            c[:k] = base
            c[:k // 2] = base * int(rng.integers(2, 9))
        elif shape == 4:                                                       # near-flat noise
            c[:k] = rng.integers(base, base + 3, k)
        else:                                                                  # steep head over a long flat tail
            c[:k] = base
            c[0] = base * k * int(rng.integers(1, 20))
            c[1] = base * k
        nz = np.nonzero(c)[0]
        msv, tot = int(nz.max()), int(c.sum())
        if tot < 2:
            continue
        vecs.append((tl, c, tot, msv))
    ok = [v for v in vecs if not is_error(checker.fse_normalize_count(*v)[0])]
    m2 = [v for v in ok if takes_m2(*v)]
    assert len(ok) > 600 and len(m2) > 100, (len(vecs), len(ok), len(m2))
    _compare_normalize(hip, checker, vecs)


def _block_with_histogram(count, rng):
    b = np.repeat(np.arange(256, dtype=np.uint8), count.astype(np.int64))
    rng.shuffle(b)
    return b


def test_unit_vector_histograms_through_the_one_shot_calls(hip, checker):
    """blocks whose histograms ARE the vectors (and x2 / x4 / x8 multiples, which keep their shape while the source grows enough for
    FSE_optimalTableLog to allow the vector's table log): FSEHIP_FSE_buildCTable_batch's header + table and FSEHIP_FSE_compress_batch's
    bytes at the vectors' table logs against the reference's one-shot call; FSE_normalizeM2 taken wherever the reference takes it"""
    rng = np.random.default_rng(9)
    cases, m2_seen = [], 0
    for tl, c, tot, msv in REFERENCE_VECTORS:
        for mult in (1, 2, 4, 8):
            cases.append((tl, _block_with_histogram(c * mult, rng)))
    for tl_req in sorted(set(t for t, _ in cases)):
        blocks = [b for t, b in cases if t == tl_req]
        size = max(len(b) for b in blocks)
        src = torch.zeros((len(blocks), size), dtype=torch.uint8, device="cuda")
        sizes = torch.tensor([len(b) for b in blocks], dtype=torch.int64, device="cuda")
        for i, b in enumerate(blocks):
            src[i, :len(b)] = torch.from_numpy(b).cuda()
        ct, hdr, hres = hip.fse_build_ctable_batch(src, table_log=tl_req, sizes=sizes)
        dst, res = hip.fse_compress_batch(src, table_log=tl_req, sizes=sizes)
        out, dres = hip.fse_decompress_batch(dst, res, size, max_log=12)
        ct_h, hdr_h, hres_h, dst_h, res_h, out_h, dres_h = (t.cpu().numpy() for t in (ct, hdr, hres, dst, res, out, dres))
        for i, b in enumerate(blocks):
            n = len(b)
            er, eout = checker.fse_compress2(b, 255, tl_req)
            assert res_h[i] == s64(er), (tl_req, n, res_h[i], er)
            mx, msv, cnt = checker.hist_count(b, 255)
            if n <= 1 or mx == n or mx == 1 or mx < (n >> 7):                  # FSE_compress_wksp stops before the table (lib/fse_compress.c:647-655)
                assert er <= 1 and hres_h[i] == s64(er), (tl_req, n)
                continue
            tl = checker.fse_optimal_tablelog(tl_req, n, msv, 2)
            m2_seen += takes_m2(tl, cnt, n, msv)
            _, norm = checker.fse_normalize_count(tl, cnt, n, msv)
            h, ehdr = checker.fse_write_ncount(512, norm, msv, tl)
            _, ect = checker.fse_build_ctable(norm, msv, tl)
            assert hres_h[i] == h and (hdr_h[i][:h] == ehdr[:h]).all(), (tl_req, n, "header")
            w = fse_ctable_u32(tl, msv)
            assert (ct_h[i].view(np.uint32)[:w] == ect[:w]).all(), (tl_req, n, "ctable")
            if er > 1:                                                         # (0: coded, but no smaller than the source -- nothing to compare)
                assert (dst_h[i][:er] == eout[:er]).all(), (tl_req, n, "bytes")
                assert dres_h[i] == n and (out_h[i][:n] == b).all(), (tl_req, n, "round trip")
    assert m2_seen >= 3, m2_seen


def test_write_and_read_ncount_bounds(hip, checker):
    """programs/fuzzer.c:367-417: the header of `i % 127` over 16 KB -- written into exactly its size, refused (and nothing written) with one
    byte less, fine with one more; read back with maxSymbolValue 128, refused with 64 (too small) and with the last byte missing -- by
    FSEHIP_FSE_writeNCount_batch / FSEHIP_FSE_readNCount_batch, and the truncated header through FSEHIP_FSE_buildDTable_batch as well"""
    buf = (np.arange(16384) % 127).astype(np.uint8)
    mx, msv, cnt = checker.hist_count(buf, 128)
    tl = checker.fse_optimal_tablelog(0, 16384, msv, 2)
    _, enorm = checker.fse_normalize_count(tl, cnt, 16384, msv)
    counts = torch.from_numpy(cnt.astype(np.int32)).cuda()[None, :].contiguous()
    totals = torch.tensor([16384], dtype=torch.int64, device="cuda")
    msvs = torch.tensor([msv], dtype=torch.int32, device="cuda")
    norms, nres = hip.fse_normalize_count_batch(counts, totals, msvs, tl)
    assert int(nres[0]) == tl and (norms[0].cpu().numpy()[:msv + 1] == enorm[:msv + 1]).all()
    full, ehdr = checker.fse_write_ncount(513, enorm, msv, tl)
    assert not is_error(full)
    caps = [513, full, full - 1, full + 1, 3, 1, 0]
    for cap in caps:
        hdr, res = hip.fse_write_ncount_batch(norms, msvs, tl, capacity=cap, stride=520)
        er, eh = checker.fse_write_ncount(cap, enorm, msv, tl)
        assert int(res[0]) == s64(er), (cap, int(res[0]), er)
        hh = hdr[0].cpu().numpy()
        if is_error(er):
            assert (hh == 0xA5).all(), cap                                      # "buffer overwrite" check of :386-388, for the whole slot
        else:
            assert (hh[:er] == eh[:er]).all() and (hh[er:] == 0xA5).all(), cap
    assert is_error(checker.fse_write_ncount(full - 1, enorm, msv, tl)[0])
    hdr, _ = hip.fse_write_ncount_batch(norms, msvs, tl, capacity=full, stride=520)
    rows = torch.cat([hdr, hdr, hdr, hdr], 0).contiguous()
    sizes = torch.tensor([full, full, full - 1, full + 7], dtype=torch.int64, device="cuda")
    limits = torch.tensor([128, 64, 128, 255], dtype=torch.int32, device="cuda")
    rn, rmsv, rtl, rres = hip.fse_read_ncount_batch(rows, sizes, limits)
    hbytes = hdr[0].cpu().numpy()
    for i, (sz, lim) in enumerate(((full, 128), (full, 64), (full - 1, 128), (full + 7, 255))):
        er, emsv, etl, en = checker.fse_read_ncount(hbytes[:sz], lim)
        assert int(rres[i]) == s64(er), (i, int(rres[i]), er)
        if not is_error(er):
            assert int(rmsv[i]) == emsv and int(rtl[i]) == etl and (rn[i].cpu().numpy()[:emsv + 1] == en[:emsv + 1]).all(), i
    assert int(rres[0]) == full and is_error(int(rres[1]) & ((1 << 64) - 1)) and is_error(int(rres[2]) & ((1 << 64) - 1))
    # the same truncation through the table builder of the decode side
    dt, dres = hip.fse_build_dtable_batch(rows[:3].contiguous(), torch.tensor([full, full - 1, 2], dtype=torch.int64, device="cuda"), max_log=12)
    for i, sz in enumerate((full, full - 1, 2)):
        er = checker.fse_read_ncount(hbytes[:sz], 255)[0]
        assert int(dres[i]) == s64(er), (i, int(dres[i]), er)


def test_read_ncount_garbage_and_every_truncation(hip, checker):
    """every prefix of real headers and random bytes through FSEHIP_FSE_readNCount_batch at several alphabet limits"""
    rng = np.random.default_rng(31)
    rows, sizes, limits = [], [], []
    for p in (2, 14, 50, 80):
        blk = checker.probagen_batch(p, 1, 4096, 77 + p)[0]
        r, out = checker.fse_compress2(blk, 255, 11)
        for cut in range(1, 40):
            rows.append(out[:64].copy()); sizes.append(cut); limits.append(255)
        for lim in (0, 1, 7, 31, 100, 254):
            rows.append(out[:64].copy()); sizes.append(64); limits.append(lim)
    for _ in range(400):
        g = rng.integers(0, 256, 64, dtype=np.uint8)
        if rng.integers(0, 2):
            g[0] = (g[0] & 0xF0) | int(rng.integers(0, 8))                     # a plausible table log
        rows.append(g); sizes.append(int(rng.integers(1, 65))); limits.append(int(rng.choice([255, 255, 60, 12])))
    hdrs = torch.from_numpy(np.stack(rows)).cuda()
    rn, rmsv, rtl, rres = hip.fse_read_ncount_batch(hdrs, torch.tensor(sizes, dtype=torch.int64, device="cuda"), torch.tensor(limits, dtype=torch.int32, device="cuda"))
    rn, rmsv, rtl, rres = rn.cpu().numpy(), rmsv.cpu().numpy(), rtl.cpu().numpy(), rres.cpu().numpy()
    good = 0
    for i in range(len(rows)):
        er, emsv, etl, en = checker.fse_read_ncount(rows[i][:sizes[i]], limits[i])
        assert rres[i] == s64(er), (i, sizes[i], limits[i], rres[i], er)
        if not is_error(er):
            good += 1
            assert rmsv[i] == emsv and rtl[i] == etl and (rn[i][:emsv + 1] == en[:emsv + 1]).all(), i
    assert good > 20
