"""development aid: cycle accounting of k_fse_decode's decoder / service waves (library built with EXTRA=-DFSE_DEC_TIMING)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C, numpy as np
from finitestateentropy_amd.api import FseHip
hip = FseHip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 7680
P = int(sys.argv[2]) if len(sys.argv) > 2 else 14
src = hip.probagen_batch(P, n, 32768, 1)
dst, res = hip.fse_compress_batch(src, 11)
for _ in range(2):
    out, dres = hip.fse_decompress_batch(dst, res, 32768, 11)
torch.cuda.synchronize()
buf = np.zeros(4096 * 8, dtype=np.uint64)
rc = hip.lib.FSEHIP_debug_decTiming(buf.ctypes.data_as(C.c_void_p))
t = buf.reshape(4096, 8)[: min(4096, n // 15)].astype(np.float64)
m = t.mean(0)
print("decoder: run %.0f cyc in %.0f phases (%.0f/phase), wait %.0f cyc in %.0f polls | service wave 0: busy %.0f cyc in %.0f rounds (%.0f/round), idle %.0f" % (
    m[0], m[2], m[0] / m[2], m[1], m[3], m[4], m[6], m[4] / max(m[6], 1), m[5]))
print("equal", bool(torch.equal(out, src)))
