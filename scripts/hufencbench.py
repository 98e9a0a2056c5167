"""development aid: Huff0 one-shot encode time per batch (hist + k_huf_cprep + k_huf_encode) for A/B runs of differently configured builds
(FSEHIP_LIB=finitestateentropy_amd/csrc/variants/x/libfsehip.so python scripts/hufencbench.py [blocks [P]]), round trip checked through the product decoder"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from finitestateentropy_amd.api import FseHip

hip = FseHip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
tag = os.environ.get("FSEHIP_LIB", "/base/x").split("/")[-2]
for P in ([int(sys.argv[2])] if len(sys.argv) > 2 else [80, 14, 2]):
    src = hip.probagen_batch(P, n, 32768, 1)
    ws = hip.huf_workspace(n, False)
    dst, res = hip.huf_compress_batch(src, 11, workspace=ws)
    run = lambda: hip.huf_compress_batch(src, 11, dst=dst, results=res, workspace=ws)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); run(); t1.record(); torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1))
    ts.sort()
    out, dres = hip.huf_decompress_batch(dst, res, 32768)
    ok = bool(torch.equal(out, src)) and bool((dres == 32768).all())
    print("%s: Huff0 encode call P%02d %d blocks: best %.3f ms, median %.3f ms, checksum %d, roundtrip ok=%s" % (tag, P, n, ts[0], ts[len(ts) // 2], int(res.sum().item()), ok))
