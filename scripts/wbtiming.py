"""development aid: phase cycle breakdown of the wave table builders (library built with EXTRA=-DFSE_WB_TIMING)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C, numpy as np
from finitestateentropy_amd.api import FseHip
hip = FseHip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30720
P = int(sys.argv[2]) if len(sys.argv) > 2 else 14
names = ["setup+clear", "marks+scan", "spread pass 1", "spread pass 2", "rank(last window)", "sums", "emit"]
src = hip.probagen_batch(P, n, 32768, 1)
def dump(tag):
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    hip.lib.FSEHIP_debug_wbTiming(buf.ctypes.data_as(C.c_void_p))
    t = buf.reshape(4096, 8)[:, :6].astype(np.float64).mean(0)
    print(tag, " ".join("%s=%.0f" % (nm, v) for nm, v in zip(names[1:], t)), "sum=%.0f" % t.sum())
for _ in range(2):
    dst, res = hip.fse_compress_batch(src, 11)
dump("cbuild:")
for _ in range(2):
    out, dres = hip.fse_decompress_batch(dst, res, 32768, 11)
dump("dbuild:")
