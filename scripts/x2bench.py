"""HUF_decompress4X_usingDTable over a batch with caller-built tables (run on the GPU box: python scripts/x2bench.py [n_blocks]).
Tables come from the compiled reference's builders (HUF_readDTableX2 / HUF_readDTableX1), as a caller of the drop-in would have them;
times the batched device call for double-symbol and single-symbol tables and checks every block's round trip."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finitestateentropy_amd.api import FseHip
from oracle.oracle import Ref

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
hip, ref = FseHip(), Ref()
for P in (14, 2):
    src = hip.probagen_batch(P, n, 32768, first_seed=1)
    cdst, cres = hip.huf_compress_batch(src, table_log=11)
    host_c, host_r = cdst.cpu().numpy(), cres.cpu().numpy()
    W = 1 + (1 << 12)
    for kind in ("x2", "x1"):
        tabs = np.zeros((n, W), np.uint32); pay = np.zeros((n, host_c.shape[1]), np.uint8); psz = np.zeros(n, np.int64)
        for b in range(n):
            c = host_c[b][:host_r[b]]
            h, dt = ref.huf_read_dtable_x2(c) if kind == "x2" else ref.huf_read_dtable_x1(c, 12)
            tabs[b, :min(len(dt), W)] = dt[:W]; pay[b, :host_r[b] - h] = c[h:]; psz[b] = host_r[b] - h
        d_t = torch.from_numpy(tabs.view(np.int32)).cuda(); d_p = torch.from_numpy(pay).cuda(); d_s = torch.from_numpy(psz).cuda()
        out, res = hip.huf_decompress4x_using_dtable_batch(d_p, d_s, d_t, 32768); torch.cuda.synchronize()
        assert bool((res == 32768).all()) and torch.equal(out, src), (P, kind)
        t0 = time.perf_counter()
        for _ in range(5):
            hip.huf_decompress4x_using_dtable_batch(d_p, d_s, d_t, 32768)
        torch.cuda.synchronize()
        dt_s = (time.perf_counter() - t0) / 5
        print("P%02d %s tables: %d blocks in %.3f ms -> %.1f GB/s (incl. the call's zero-filled output allocation)" % (P, kind, n, dt_s * 1e3, n * 32768 / dt_s / 1e9))
