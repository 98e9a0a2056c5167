"""development aid: time the Huff0 prepare kernels of a given library build (FSEHIP_LIB) on P14 and P02"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from finitestateentropy_amd.api import FseHip
hip = FseHip()
N = 100000
for P in (14, 2):
    src = hip.probagen_batch(P, N, 32768, 1)
    dst, res = hip.huf_compress_batch(src, table_log=11)
    out, dres = hip.huf_decompress_batch(dst, res, 32768)
    torch.cuda.synchronize()
    assert torch.equal(out, src)
    hip.lib.FSEHIP_probe_begin()
    for _ in range(3):
        hip.huf_compress_batch(src, table_log=11, dst=dst, results=res)
        hip.huf_decompress_batch(dst, res, 32768, dst=out, results=dres)
    torch.cuda.synchronize()
    ms = (C.c_double * 16)(); n = (C.c_uint * 16)()
    hip.lib.FSEHIP_probe_collect(ms, n)
    print(os.environ.get("FSEHIP_LIB", "default"), "P%02d" % P, "cprep %.3f ms  dprep %.3f ms" % (ms[5] / 3, ms[7] / 3))
