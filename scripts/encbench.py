"""development aid: FSE one-shot encode time per batch (hist + cprep + lists + k_fse_encode_wave / k_fse_encode) for A/B runs of differently
configured builds (FSEHIP_LIB=finitestateentropy_amd/csrc/variants/x/libfsehip.so python scripts/encbench.py [blocks]): P14 / P80 / P02 / P50 / P20,
with the round trip checked through the product decoder"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from finitestateentropy_amd.api import FseHip

hip = FseHip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
tag = os.environ.get("FSEHIP_LIB", "/base/x").split("/")[-2]
for P in (14, 80, 2, 50, 20):
    src = hip.probagen_batch(P, n, 32768, 1)
    ws = hip.fse_workspace(n, 11)
    dst = torch.empty((n, 33548), dtype=torch.uint8, device="cuda"); res = torch.empty(n, dtype=torch.int64, device="cuda")
    run = lambda: hip.fse_compress_batch(src, 11, dst=dst, results=res, workspace=ws)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); run(); t1.record(); torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1))
    ts.sort()
    out, dres = hip.fse_decompress_batch(dst, res, 32768, 12)
    ok = bool(torch.equal(out, src)) and bool((dres == 32768).all())
    print("%s: FSE encode call P%02d %d blocks: best %.3f ms, median %.3f ms, checksum %d, roundtrip ok=%s" % (tag, P, n, ts[0], ts[len(ts) // 2], int(res.sum().item()), ok))
