"""development aid: the 16-bit coder's two batch calls timed alone (25k blocks of 16,384 symbols, the bench's corpus): PYTHONPATH=. python scripts/u16bench.py [tag]
(A/B of builds: make B=variants/x EXTRA=-D..., FSEHIP_LIB=...; EXPERIMENTS.md section 4)"""
import numpy as np, torch, sys
from finitestateentropy_amd import api
hip = api.FseHip()
nb, nsym = 25000, 16384
rng = np.random.default_rng(16)
table = np.zeros(4096, np.uint16)
remaining, pos, val = 4096, 0, 240
while remaining:
    k = int(remaining * 0.08) + 1
    table[pos:pos + k] = val; pos += k; remaining -= k
    val = val + 1 if val + 1 < 286 else 1
host = table[rng.integers(0, 4096, (256, nsym))]
src = torch.from_numpy(host.view(np.int16)).cuda().repeat((nb + 255) // 256, 1)[:nb].contiguous()
cdst, cres = hip.fse_compress_u16_batch(src)
out, dres = hip.fse_decompress_u16_batch(cdst, cres, nsym)
torch.cuda.synchronize()
def t(f, n=10):
    best = 1e9
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best
print(sys.argv[1] if len(sys.argv) > 1 else "base", "compress %.3f ms  decompress %.3f ms" % (t(lambda: hip.fse_compress_u16_batch(src, dst=cdst, results=cres)), t(lambda: hip.fse_decompress_u16_batch(cdst, cres, nsym, dst=out, results=dres))), flush=True)
