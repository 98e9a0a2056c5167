"""development aid: very many small blocks (beyond one workspace chunk of 131072 blocks) through the one-shot calls"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from finitestateentropy_amd.api import FseHip
from oracle.oracle import Checker
hip = FseHip(); o = Checker()
for n, size in ((300000, 100), (270001, 257), (150000, 2048)):
    src = hip.probagen_mixed((2, 14, 80), n, size)
    fd, fr = hip.fse_compress_batch(src, 11)
    hd, hr = hip.huf_compress_batch(src)
    torch.cuda.synchronize()
    idx = np.linspace(0, n - 1, 400).astype(np.int64)
    host = src[idx].cpu().numpy(); frh = fr[idx].cpu().numpy(); hrh = hr[idx].cpu().numpy(); fdh = fd[idx].cpu().numpy(); hdh = hd[idx].cpu().numpy()
    for k in range(len(idx)):
        rr, rout = o.fse_compress2(host[k], 255, 11)
        assert int(frh[k]) == rr and (rr <= 1 or (fdh[k][:rr] == rout[:rr]).all()), ("fse", n, size, idx[k])
        rr, rout = o.huf_compress2(host[k], 255, 11)
        assert int(hrh[k]) == rr and (rr <= 1 or (hdh[k][:rr] == rout[:rr]).all()), ("huf", n, size, idx[k])
    okf = fr > 1; okh = hr > 1
    out, dr = hip.fse_decompress_batch(fd[okf], fr[okf], size)
    hout, hdr = hip.huf_decompress_batch(hd[okh], hr[okh], size)
    torch.cuda.synchronize()
    assert torch.equal(out, src[okf]) and torch.equal(hout, src[okh])
    print("ok", n, size, int(okf.sum()), int(okh.sum()))
