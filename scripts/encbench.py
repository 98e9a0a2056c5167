"""tiny driver: FSE encode only, for PMC profiling"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from finitestateentropy_amd.api import FseHip
hip = FseHip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
P = int(sys.argv[2]) if len(sys.argv) > 2 else 14
src = hip.probagen_batch(P, n, 32768, 1)
for _ in range(3):
    dst, res = hip.fse_compress_batch(src, 11)
torch.cuda.synchronize()
print("ok", int(res.sum().item()))
