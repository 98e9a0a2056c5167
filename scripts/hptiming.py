"""development aid: phase cycle accounting of k_huf_cprep / k_huf_dprep (library built with EXTRA=-DHP_TIMING)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from finitestateentropy_amd.api import FseHip
hip = FseHip()
for P in (14, 2):
    src = hip.probagen_batch(P, 16384, 32768, 1)
    for tl in (11,):
        dst, res = hip.huf_compress_batch(src, table_log=tl)
        out, dres = hip.huf_decompress_batch(dst, res, 32768)
        torch.cuda.synchronize()
        assert torch.equal(out, src)
        buf = np.zeros((2, 2048, 12), np.uint64)
        hip.lib.FSEHIP_debug_hpTiming(buf.ctypes.data_as(C.c_void_p))
        for which, name in ((0, "cprep A sort|B merge|C chase|D repair|E1 codes|E2 wstats|E3 stab|F wenc|G hdr"), (1, "dprep A parse|B stab|C wdec|D fill")):
            m = buf[which].astype(np.float64).mean(axis=0)
            print("P%02d" % P, name, " ".join("%8.0f" % x for x in m[:9]), " total %.0f" % m[:9].sum())
