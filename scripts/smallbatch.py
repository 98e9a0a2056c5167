"""development aid: what a call costs on small batches (launch overheads, empty class launches): FSE / Huff0 one-shot encode and decode calls over
n = 64 .. 16384 blocks of 32 KB (P14), best of 20, microseconds per call (FSEHIP_LIB=... for A/B builds)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from finitestateentropy_amd.api import FseHip

hip = FseHip()
tag = os.environ.get("FSEHIP_LIB", "/base/x").split("/")[-2]

def best(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); fn(); t1.record(); torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1) * 1000.0)
    return min(ts)

for n in (64, 512, 2048, 16384):
    src = hip.probagen_batch(14, n, 32768, 1)
    ws = hip.fse_workspace(n, 11)
    dst = torch.empty((n, 33548), dtype=torch.uint8, device="cuda"); res = torch.empty(n, dtype=torch.int64, device="cuda")
    enc = lambda: hip.fse_compress_batch(src, 11, dst=dst, results=res, workspace=ws)
    e = best(enc)
    out, dres = hip.fse_decompress_batch(dst, res, 32768, 12)
    dec = lambda: hip.fse_decompress_batch(dst, res, 32768, 12, dst=out, results=dres)
    d = best(dec)
    hd, hres = hip.huf_compress_batch(src)
    he = best(lambda: hip.huf_compress_batch(src, dst=hd, results=hres))
    ho, hdres = hip.huf_decompress_batch(hd, hres, 32768)
    hdd = best(lambda: hip.huf_decompress_batch(hd, hres, 32768, dst=ho, results=hdres))
    print("%s: %6d blocks: FSE encode %8.1f us  decode %8.1f us | Huff0 encode %8.1f us  decode %8.1f us" % (tag, n, e, d, he, hdd))
