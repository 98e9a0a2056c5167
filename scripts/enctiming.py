"""development aid: stage cycle breakdown of k_fse_encode_wave (library built with EXTRA=-DFSE_ENC_TIMING)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C, numpy as np
from finitestateentropy_amd.api import FseHip
hip = FseHip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
for P in [int(x) for x in sys.argv[2:]] or [14]:
    src = hip.probagen_batch(P, n, 32768, 1)
    for _ in range(2):
        dst, res = hip.fse_compress_batch(src, 11)
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    hip.lib.FSEHIP_debug_encTiming(buf.ctypes.data_as(C.c_void_p))
    t = buf.reshape(4096, 8).astype(np.float64)
    m = t.mean(0)
    print("P%d cycles/block: stage %.0f  warm+count %.0f  repair %.0f (rounds mean %.2f max %.0f)  prefix+emit %.0f  total %.0f" % (
        P, m[0], m[1], m[2], m[4], t[:, 4].max(), m[3], m[:4].sum()))
    print("    lanes repaired in round 1: mean %.2f" % m[5])
