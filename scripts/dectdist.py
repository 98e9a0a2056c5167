import sys, os
sys.path.insert(0, "/root/repo")
import torch, ctypes as C, numpy as np
from finitestateentropy_amd.api import FseHip
hip = FseHip()
n = 16384
src = hip.probagen_batch(14, n, 32768, 1)
dst, res = hip.fse_compress_batch(src, 11)
for _ in range(2):
    out, dres = hip.fse_decompress_batch(dst, res, 32768, 11)
torch.cuda.synchronize()
buf = np.zeros(4096 * 8, dtype=np.uint64)
hip.lib.FSEHIP_debug_decTiming(buf.ctypes.data_as(C.c_void_p))
t = buf.reshape(4096, 8)[: n // 16].astype(np.float64)
run = t[:, 0] / np.maximum(t[:, 2], 1)
print("per-phase run ticks: min %.0f p10 %.0f median %.0f p90 %.0f max %.0f" % (run.min(), np.percentile(run, 10), np.median(run), np.percentile(run, 90), run.max()))
h, e = np.histogram(run, bins=12)
print(list(zip(e[:-1].astype(int), h)))
