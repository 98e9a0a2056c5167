"""The 16-bit-symbol coder against committed vectors of the reference (tests/golden/golden_u16_v1.npz, made by
tests/golden/make_golden_u16.py from lib/fseU16.c): on the CPU the fixture is re-derived from the compiled reference when it is
present (so the fixture cannot rot); on the GPU the device path must reproduce it whether or not oracle/_ref travelled."""
import os

import numpy as np
import pytest

from oracle.oracle import Ref, is_error

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_u16_v1.npz"))


def test_u16_golden_is_what_the_reference_produces(gold):
    if not Ref.available():
        pytest.skip("oracle/_ref/libfse_ref.so not built (needs the reference tree)")
    r = Ref()
    meta = gold["meta"]
    assert len(meta) >= 20
    for i, (n, tl, cs, mx, msv, ds, bs) in enumerate(meta.tolist()):
        src = gold["src_%d" % i]
        rc, comp = r.fse_compress_u16(src, 0, int(tl))
        assert rc == cs, i
        if 1 < cs < (1 << 62):
            assert (comp[:cs] == gold["comp_%d" % i]).all(), i
            assert r.fse_decompress_u16(gold["bad_%d" % i], int(n))[0] == bs, i
        m, cnt, sv = r.fse_count_u16(src, 286)
        assert m == mx and sv == msv and (cnt[:287] == gold["count_%d" % i]).all(), i


@pytest.mark.gpu
def test_u16_device_reproduces_golden(hip, gold):
    for i, (n, tl, cs, mx, msv, ds, bs) in enumerate(gold["meta"].tolist()):
        src = gold["src_%d" % i]
        rc, comp = hip.fse_compress_u16(src, 0, int(tl))
        assert rc == cs, (i, rc, cs)
        m, cnt, sv = hip.fse_count_u16(src, 286)
        assert m == mx and sv == msv and (cnt[:287] == gold["count_%d" % i]).all(), i
        if 1 < cs < (1 << 62):
            assert (comp[:cs] == gold["comp_%d" % i]).all(), i
            d, dec = hip.fse_decompress_u16(gold["comp_%d" % i], int(n))
            assert d == ds == n and (dec[:n] == src).all(), i
            assert hip.fse_decompress_u16(gold["bad_%d" % i], int(n))[0] == bs, i
