"""development aid: cycle accounting of k_huf_decode's decoder wave (library built with EXTRA=-DHD_TIMING)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C, numpy as np
from finitestateentropy_amd.api import FseHip
hip = FseHip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
P = int(sys.argv[2]) if len(sys.argv) > 2 else 14
src = hip.probagen_batch(P, n, 32768, 1)
dst, res = hip.huf_compress_batch(src, 11)
for _ in range(2):
    out, dres = hip.huf_decompress_batch(dst, res, 32768)
torch.cuda.synchronize()
buf = np.zeros(4096 * 4, dtype=np.uint64)
hip.lib.FSEHIP_debug_hdTiming(buf.ctypes.data_as(C.c_void_p))
t = buf.reshape(4096, 4)[: min(4096, n // 12)].astype(np.float64)
m = t.mean(0)
print("decoder: run %.0f cyc in %.0f phases (%.0f/phase), wait %.0f cyc in %.0f polls" % (m[0], m[2], m[0] / m[2], m[1], m[3]))
print("equal", bool(torch.equal(out, src)))
