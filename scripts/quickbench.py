"""Quick per-kernel timing on one GPU (development aid; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from finitestateentropy_amd.api import FseHip

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(iters):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); fn(); t1.record(); torch.cuda.synchronize()
        best = min(best, t0.elapsed_time(t1) / 1e3)
    return best

def main():
    hip = FseHip()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    for P in (14, 80, 2):
        src = hip.probagen_batch(P, n, 32768, 1)
        nbytes = n * 32768
        t = timeit(lambda: hip.hist_count_batch(src))
        print("P%02d hist        %8.2f GB/s" % (P, nbytes / t / 1e9))
        ws = hip.fse_workspace(n, 11)
        dst = torch.empty((n, 33548), dtype=torch.uint8, device="cuda"); res = torch.empty(n, dtype=torch.int64, device="cuda")
        t = timeit(lambda: hip.fse_compress_batch(src, 11, dst=dst, results=res, workspace=ws))
        csum = int(res.sum().item())
        print("P%02d fse enc     %8.2f GB/s  (ratio %.3f)" % (P, nbytes / t / 1e9, nbytes / csum))
        wsd = hip.fse_workspace(n, 11, True)
        out = torch.empty((n, 32768), dtype=torch.uint8, device="cuda"); dres = torch.empty(n, dtype=torch.int64, device="cuda")
        t = timeit(lambda: hip.fse_decompress_batch(dst, res, 32768, 11, dst=out, results=dres, workspace=wsd))
        print("P%02d fse dec     %8.2f GB/s  ok=%s" % (P, nbytes / t / 1e9, bool(torch.equal(out, src))))
        if hasattr(hip, "huf_compress_batch"):
            hws = hip.huf_workspace(n)
            t = timeit(lambda: hip.huf_compress_batch(src, 11, dst=dst, results=res, workspace=hws))
            csum = int(res.sum().item())
            print("P%02d huf enc     %8.2f GB/s  (ratio %.3f)" % (P, nbytes / t / 1e9, nbytes / csum))
            hwd = hip.huf_workspace(n, True)
            t = timeit(lambda: hip.huf_decompress_batch(dst, res, 32768, dst=out, results=dres, workspace=hwd))
            print("P%02d huf dec     %8.2f GB/s  ok=%s" % (P, nbytes / t / 1e9, bool(torch.equal(out, src))))

if __name__ == "__main__":
    main()
