import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from finitestateentropy_amd.api import FseHip
hip = FseHip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 15360
src = hip.probagen_batch(14, n, 32768, 1)
dst, res = hip.fse_compress_batch(src, 11)
ws = hip.fse_workspace(n, 11, True)
out = torch.empty((n, 32768), dtype=torch.uint8, device="cuda"); dres = torch.empty(n, dtype=torch.int64, device="cuda")
for _ in range(2): hip.fse_decompress_batch(dst, res, 32768, 11, dst=out, results=dres, workspace=ws)
torch.cuda.synchronize()
hip.lib.FSEHIP_probe_begin()
for _ in range(3): hip.fse_decompress_batch(dst, res, 32768, 11, dst=out, results=dres, workspace=ws)
ms = (C.c_double * 16)(); ln = (C.c_uint * 16)()
hip.lib.FSEHIP_probe_collect(ms, ln)
print("n", n, "decode kernel ms/launch", ms[4] / ln[4], "launches", ln[4], "dprep", ms[3] / ln[3], "equal", bool(torch.equal(out, src)))
