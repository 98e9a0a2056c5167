"""development aid: FSE_compress over a ragged batch (per-block sizes 12,000 ... 32,768 bytes) against the uniform 32 KB batch: PYTHONPATH=. python scripts/raggedbench.py
(the encoders' per-block split, csrc/internal.h launch_fse_encode_auto; EXPERIMENTS.md section 2)"""
import numpy as np, torch
from finitestateentropy_amd import api
hip = api.FseHip()
nb = 20000
src = hip.probagen_batch(14, nb, 32768, 1)
rng = np.random.default_rng(3)
sizes = torch.from_numpy(rng.integers(12000, 32769, nb).astype(np.int64)).cuda()
sizes[:8] = torch.tensor([0, 1, 5, 100, 2047, 2048, 2049, 32768])
def t(f, n=5):
    best = 1e9
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best
cd, cr = hip.fse_compress_batch(src, sizes=sizes)
torch.cuda.synchronize()
ms = t(lambda: hip.fse_compress_batch(src, sizes=sizes, dst=cd, results=cr))
print("ragged FSE_compress batch: %d blocks of 12000..32768 bytes: %.3f ms = %.1f GB/s" % (nb, ms, float(sizes.sum()) / ms / 1e6))
cd2, cr2 = hip.fse_compress_batch(src)
ms = t(lambda: hip.fse_compress_batch(src, dst=cd2, results=cr2))
print("uniform 32 KB: %.3f ms = %.1f GB/s" % (ms, nb * 32768 / ms / 1e6))
