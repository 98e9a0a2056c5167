"""Generate tests/golden/golden_v1.npz from the UNMODIFIED reference (oracle/_ref/libfse_ref.so).

Run in the build container (where /root/reference exists):   python tests/golden/make_golden.py
The reference cannot travel to the GPU box, these vectors can.  Every vector is an output of the
reference's own functions (FSE_compress2, HUF_compress2, FSE_normalizeCount, FSE_writeNCount,
FSE_buildCTable, FSE_buildDTable, HUF_buildCTable, HUF_writeCTable, HUF_readDTableX1, HIST_count; frames from its
command-line tool, programs/fileio.c) on
probagen blocks (programs/probaGenerator.c restated in oracle/fse_oracle.c, checked against SURVEY
Appendix B source hashes).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle, Ref, is_error  # noqa: E402


def main():
    o, r = Oracle(), Ref()
    out = {}
    cases = []          # (P, seed, n)
    for P in (2, 14, 80):
        for seed in (1, 2, 3):
            cases.append((P, seed, 32768))
        cases.append((P, 7, 4096))
        cases.append((P, 9, 4097))
    cases += [(20, 5, 1000), (90, 5, 777), (1, 5, 2048), (50, 11, 32767)]
    meta = []
    for i, (P, seed, n) in enumerate(cases):
        blk = o.probagen_batch(P, 1, n, seed)[0]
        mx, msv, cnt = r.hist_count(blk)
        fc, fbuf = r.fse_compress2(blk, 255, 11)
        fc12, fbuf12 = r.fse_compress2(blk, 255, 12)
        hc, hbuf = r.huf_compress2(blk, 255, 11)
        meta.append((P, seed, n, o.xxh64(blk), mx, msv, fc, o.xxh64(fbuf[:fc]), fc12, o.xxh64(fbuf12[:fc12]), hc, o.xxh64(hbuf[:hc])))
        out["count_%d" % i] = cnt
        if n <= 4097 or (seed == 1 and n == 32768):
            out["fse_%d" % i] = fbuf[:fc].copy()
            out["huf_%d" % i] = hbuf[:hc].copy()
        # table-level vectors for the first block of each distribution
        if seed == 1 and n == 32768:
            tl = r.fse_optimal_tablelog(11, n, msv, 2)
            _, norm = r.fse_normalize_count(tl, cnt, n, msv)
            hs, hdr = r.fse_write_ncount(512, norm, msv, tl)
            _, ct = r.fse_build_ctable(norm, msv, tl)
            _, dt = r.fse_build_dtable(norm, msv, tl)
            hl = r.fse_optimal_tablelog(11, n, msv, 1)
            mb, celt = r.huf_build_ctable(cnt, msv, hl)
            whs, whdr = r.huf_write_ctable(256, celt, msv, mb)
            _, hdt = r.huf_read_dtable_x1(whdr[:whs], 11)
            cs, payload = r.fse_compress_using_ctable(blk, ct)
            out["tl_%d" % i] = np.array([tl, msv, hs, mb, whs, cs], dtype=np.int64)
            out["norm_%d" % i] = norm
            out["ncount_%d" % i] = hdr[:hs].copy()
            # CTable: deltaFindState of absent symbols is uninitialised in the reference -> mask it
            ct = ct.copy()
            tt = 1 + (1 << (tl - 1))
            for s in range(msv + 1):
                if norm[s] == 0:
                    ct[tt + 2 * s] = 0
            out["ctable_%d" % i] = ct
            out["dtable_%d" % i] = dt
            out["celt_%d" % i] = (celt[:msv + 1] & 0x00FFFFFF)
            out["hufhdr_%d" % i] = whdr[:whs].copy()
            out["hufdt_%d" % i] = hdt[:1 + (1 << mb)]
    # .fse frames written by the reference's command-line tool (oracle/_ref/fse_cli): 2.5 blocks of P14 + an RLE block + noise
    import subprocess, tempfile
    rng = np.random.default_rng(1)
    fsrc = np.concatenate([o.probagen_batch(14, 1, 81920, 21)[0], np.full(32768, 3, np.uint8), rng.integers(0, 256, 5000, dtype=np.uint8)])
    cli = os.path.join(ROOT, "oracle", "_ref", "fse_cli")
    with tempfile.TemporaryDirectory() as d:
        fsrc.tofile(os.path.join(d, "in"))
        for key, flag in (("frame_fse", "-fqq"), ("frame_huf", "-fqqh")):
            subprocess.run([cli, flag, os.path.join(d, "in"), os.path.join(d, key)], check=True, capture_output=True)
            out[key] = np.fromfile(os.path.join(d, key), dtype=np.uint8)
    out["frame_src"] = fsrc
    out["meta"] = np.array(meta, dtype=np.uint64)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(cases), "cases")


if __name__ == "__main__":
    main()
