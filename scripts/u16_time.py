"""development aid: the 16-bit-symbol coder's kernels under rocprofv3 --kernel-trace --stats (25k blocks of 16384 symbols)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from finitestateentropy_amd.api import FseHip
hip = FseHip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
rng = np.random.default_rng(16)
table = np.zeros(4096, np.uint16); remaining, pos, val = 4096, 0, 240
while remaining:
    k = int(remaining * 0.08) + 1; table[pos:pos + k] = val; pos += k; remaining -= k; val = val + 1 if val + 1 < 286 else 1
host = table[rng.integers(0, 4096, (256, 16384))]
src = torch.from_numpy(host.view(np.int16)).cuda().repeat((n + 255) // 256, 1)[:n].contiguous()
for _ in range(3):
    c, r = hip.fse_compress_u16_batch(src)
    o, d = hip.fse_decompress_u16_batch(c, r, 16384)
torch.cuda.synchronize()
print("ok", bool(torch.equal(o, src)))
