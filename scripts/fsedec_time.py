"""development aid: time FSEHIP_FSE_decompress_batch alone (prepare + decode), 100k blocks: python scripts/fsedec_time.py [P ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from finitestateentropy_amd.api import FseHip
hip = FseHip()
n = 100000
for P in [int(x) for x in sys.argv[1:]] or [14]:
    src = hip.probagen_batch(P, n, 32768, 1)
    dst, res = hip.fse_compress_batch(src, 11)
    for _ in range(2): out, dres = hip.fse_decompress_batch(dst, res, 32768)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): out, dres = hip.fse_decompress_batch(dst, res, 32768)
    e1.record(); torch.cuda.synchronize()
    print("%s P%02d decompress (prepare + decode) %.3f ms   exact %s" % (os.environ.get("FSEHIP_LIB", "default"), P, e0.elapsed_time(e1) / 5, bool(torch.equal(out, src))))
