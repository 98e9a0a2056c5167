"""development aid: repair statistics / phase cycles of k_huf_decode_par (library built with EXTRA=-DHPAR_STATS)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from finitestateentropy_amd.api import FseHip
hip = FseHip()
for P in (14, 80, 2):
    src = hip.probagen_batch(P, 16384, 32768, 1)
    dst, res = hip.huf_compress_batch(src, table_log=11)
    out, dres = hip.huf_decompress_batch(dst, res, 32768)
    torch.cuda.synchronize()
    h = dst.cpu().numpy(); r = res.cpu().numpy()
    assert torch.equal(out, src)
    buf = np.zeros((4096, 8), np.uint64)
    hip.lib.FSEHIP_debug_hparStats(buf.ctypes.data_as(C.c_void_p))
    m = buf.astype(np.float64).mean(axis=0)
    print("P%02d compressed size mean %.0f max %d" % (P, r.mean(), r.max()))
    print("P%02d per block (4 streams): repair rounds %.2f  bad links %.2f  cycles: stage %.0f  pass1 %.0f  repair %.0f  verdict+pass2 %.0f" % (P, m[0], m[1], m[2], m[3], m[4], m[5]))
