"""development aid: time the FSE prepare kernels of a given library build (FSEHIP_LIB)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from finitestateentropy_amd.api import FseHip
hip = FseHip()
N = 100000
for P, tl in ((14, 11), (2, 11), (14, 12)):
    src = hip.probagen_batch(P, N, 32768, 1)
    dst, res = hip.fse_compress_batch(src, table_log=tl)
    out, dres = hip.fse_decompress_batch(dst, res, 32768, max_log=12)
    torch.cuda.synchronize()
    assert torch.equal(out, src)
    hip.lib.FSEHIP_probe_begin()
    for _ in range(3):
        hip.fse_compress_batch(src, table_log=tl, dst=dst, results=res)
        hip.fse_decompress_batch(dst, res, 32768, max_log=12, dst=out, results=dres)
    torch.cuda.synchronize()
    ms = (C.c_double * 16)(); n = (C.c_uint * 16)()
    hip.lib.FSEHIP_probe_collect(ms, n)
    print(os.environ.get("FSEHIP_LIB", "default"), "P%02d tl%d" % (P, tl), "hist %.3f cprep %.3f dprep %.3f enc %.3f dec %.3f" % (ms[0] / 3, ms[1] / 3, ms[3] / 3, (ms[2] + ms[9]) / 3, ms[4] / 3))
