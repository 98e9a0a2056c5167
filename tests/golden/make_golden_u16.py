"""Generate tests/golden/golden_u16_v1.npz from the UNMODIFIED reference (oracle/_ref/libfse_ref.so, which includes lib/fseU16.c).

Run in the build container (where /root/reference exists):   python tests/golden/make_golden_u16.py
Every vector is an output of the reference's FSE_countU16 / FSE_compressU16 / FSE_decompressU16 on the committed inputs (the
inputs are stored too: numpy's generators are not a format)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.oracle import Ref  # noqa: E402
from test_gpu_u16 import u16_block, KINDS  # noqa: E402


def main():
    r = Ref()
    rng = np.random.default_rng(20260924)
    out, meta = {}, []
    i = 0
    for kind in KINDS + ("rle",):
        for n, tl in ((4097, 0), (7001, 9), (3, 0)):
            src = u16_block(rng, n, kind)
            cs, comp = r.fse_compress_u16(src, 0, tl)
            mx, cnt, msv = r.fse_count_u16(src, 286)
            out["src_%d" % i] = src
            out["comp_%d" % i] = comp[:cs].copy() if 1 < cs < (1 << 62) else np.zeros(0, np.uint8)
            out["count_%d" % i] = cnt[:287].copy()
            ds = 0
            if 1 < cs < (1 << 62):
                ds, dec = r.fse_decompress_u16(comp[:cs], n)
                assert ds == n and (dec[:n] == src).all()
                bad = comp[:cs].copy()
                bad[cs - 2] ^= 0x5A                       # one damaged payload byte: the reference's verdict on it
                bs, _ = r.fse_decompress_u16(bad, n)
                out["bad_%d" % i] = bad
            else:
                bs = 0
            meta.append((n, tl, cs, mx, msv, ds, bs))
            i += 1
    out["meta"] = np.array(meta, dtype=np.uint64)
    path = os.path.join(ROOT, "tests", "golden", "golden_u16_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", i, "cases")


if __name__ == "__main__":
    main()
