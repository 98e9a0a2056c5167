"""development aid: blocks far beyond 32 KB (1 MB .. 64 MB) through every one-shot call, bytes against the compiled reference"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from finitestateentropy_amd.api import FseHip, fse_compress_bound
from oracle.oracle import Checker, Ref, is_error
hip = FseHip(); o = Checker(); ref = Ref() if Ref.available() else None
for size in (1 << 20, (1 << 22) + 12345, 1 << 26):
    for P in (14, 80, 2):
        src = hip.probagen_batch(P, 3, size, 7)
        host = src.cpu().numpy()
        fd, fr = hip.fse_compress_batch(src, 11); out, dr = hip.fse_decompress_batch(fd, fr, size)
        torch.cuda.synchronize()
        assert torch.equal(out, src) and bool((dr == size).all()), ("fse", size, P)
        rr, rout = o.fse_compress2(host[1], 255, 11)
        assert int(fr[1]) == rr and (fd[1, :rr].cpu().numpy() == rout[:rr]).all(), ("fse bytes", size, P)
        if size <= 128 * 1024 or True:
            hd, hr = hip.huf_compress_batch(src)
            torch.cuda.synchronize()
            rr, rout = o.huf_compress2(host[1], 255, 11)
            rr_s = rr - (1 << 64) if rr >= (1 << 63) else rr
            assert int(hr[1]) == rr_s, ("huf size", size, P, int(hr[1]), rr_s)    # (blocks above 128 KB: srcSize_wrong, huf_compress.c:658)
            if rr_s > 1:
                assert (hd[1, :rr].cpu().numpy() == rout[:rr]).all()
                hout, hdr = hip.huf_decompress_batch(hd, hr, size)
                torch.cuda.synchronize()
                assert torch.equal(hout, src)
        if ref is not None:
            s16 = (host[0][: min(size, 1 << 22)].view(np.uint16) % 287).astype(np.uint16)
            r, c = hip.fse_compress_u16(s16)
            rr, rc = ref.fse_compress_u16(s16)
            assert r == rr and (rr <= 1 or is_error(rr) or (c[:rr] == rc[:rr]).all()), ("u16", size, P, r, rr)
            if rr > 1 and not is_error(rr):
                d, dec = hip.fse_decompress_u16(c[:rr], s16.size)
                assert d == s16.size and (dec == s16).all()
        print("ok", size, P, int(fr[1]))
