"""development aid: Huff0 one-shot decode time per batch (dprep + k_huf_decode_par [+ serial]) for A/B runs of differently configured builds
(FSEHIP_LIB=finitestateentropy_amd/csrc/variants/x/libfsehip.so python scripts/hufdecbench.py [blocks [P]]); HUFDEC_NOCHECK=1 skips the round-trip check
(builds that leave the output unwritten on purpose)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from finitestateentropy_amd.api import FseHip

hip = FseHip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
for P in ([int(sys.argv[2])] if len(sys.argv) > 2 else [80, 14, 2]):
    src = hip.probagen_batch(P, n, 32768, 1)
    dst, res = hip.huf_compress_batch(src, 11)
    ws = hip.huf_workspace(n, True)
    out = torch.empty((n, 32768), dtype=torch.uint8, device="cuda"); dres = torch.empty(n, dtype=torch.int64, device="cuda")
    run = lambda: hip.huf_decompress_batch(dst, res, 32768, dst=out, results=dres, workspace=ws)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); run(); t1.record(); torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1))
    ts.sort()
    ok = "unchecked" if os.environ.get("HUFDEC_NOCHECK") else str(bool(torch.equal(out, src)) and bool((dres == 32768).all()))
    print("%s: Huff0 P%02d %d blocks: decode call best %.3f ms, median %.3f ms, ok=%s" % (os.environ.get("FSEHIP_LIB", "base").split("/")[-2] if os.environ.get("FSEHIP_LIB") else "base", P, n, ts[0], ts[len(ts) // 2], ok))
