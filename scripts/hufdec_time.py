"""development aid: time FSEHIP_HUF_decompress_batch alone (no parity check -- for experiment builds): P14 / P80 / P02, 100k blocks"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from finitestateentropy_amd.api import FseHip
hip = FseHip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
for P in (14, 80, 2):
    src = hip.probagen_batch(P, n, 32768, 1)
    dst, res = hip.huf_compress_batch(src, table_log=11)
    for _ in range(2): out, dres = hip.huf_decompress_batch(dst, res, 32768)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): out, dres = hip.huf_decompress_batch(dst, res, 32768)
    e1.record(); torch.cuda.synchronize()
    print("P%02d decompress (prepare + decode) %.3f ms   exact %s" % (P, e0.elapsed_time(e1) / 5, bool(torch.equal(out, src))))
