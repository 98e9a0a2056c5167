"""development aid: FSE one-shot decode time per batch (dparse + dbuild + k_fse_decode) for A/B runs of differently configured builds
(FSEHIP_LIB=finitestateentropy_amd/csrc/variants/x/libfsehip.so python scripts/decbench.py [blocks [P [tableLog]]])"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from finitestateentropy_amd.api import FseHip

hip = FseHip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 14
tl = int(sys.argv[3]) if len(sys.argv) > 3 else 11
src = hip.probagen_batch(P, n, 32768, 1)
dst, res = hip.fse_compress_batch(src, tl)
ws = hip.fse_workspace(n, max(tl, 11), True)
out = torch.empty((n, 32768), dtype=torch.uint8, device="cuda"); dres = torch.empty(n, dtype=torch.int64, device="cuda")
run = lambda: hip.fse_decompress_batch(dst, res, 32768, max(tl, 11), dst=out, results=dres, workspace=ws)
for _ in range(3):
    run()
torch.cuda.synchronize()
ts = []
for _ in range(12):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); run(); t1.record(); torch.cuda.synchronize()
    ts.append(t0.elapsed_time(t1))
ts.sort()
print("%s: P%02d tl%d %d blocks: decode best %.3f ms, median %.3f ms, ok=%s" % (os.environ.get("FSEHIP_LIB", "base"), P, tl, n, ts[0], ts[len(ts) // 2], bool(torch.equal(out, src)) and bool((dres == 32768).all())))
if os.environ.get("DECBENCH_TIMING"):
    import ctypes as C
    L = hip.lib
    L.FSEHIP_debug_decodeTiming(1, None)
    run(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    L.FSEHIP_debug_decodeTiming(0, buf)
    t = [int(x) for x in buf]
    rounds = max(t[2], 1)
    print("  decoder wave: %.0f cycles per productive round (%.1f per iteration, %.1f inside the phase), waiting rounds %.1f%% of the time; "
          "service wave 0 busy %.0f%%" % (t[0] / rounds, t[0] / rounds / 16, t[7] / rounds / 16, 100.0 * t[1] / max(t[0] + t[1], 1), 100.0 * t[5] / max(t[5] + t[6], 1)))
