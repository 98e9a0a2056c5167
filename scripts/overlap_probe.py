"""development aid: how much does piecewise pipelining over several streams buy?  Splits a batch into K pieces, runs each piece's
compress (or decompress) call on one of NS streams, optionally staggered (piece k starts after piece k-1's first kernel would have
run: approximated by an event recorded after piece k-1's call was enqueued on a helper chain), and compares wall time with one call.
usage: python scripts/overlap_probe.py [fse|huf] [P]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from finitestateentropy_amd.api import FseHip
hip = FseHip()
codec = sys.argv[1] if len(sys.argv) > 1 else "fse"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 14
n = 100000
src = hip.probagen_batch(P, n, 32768, 1)
comp = hip.fse_compress_batch if codec == "fse" else hip.huf_compress_batch
if codec == "fse":
    decomp = lambda c, r, dst=None, results=None, workspace=None: hip.fse_decompress_batch(c, r, 32768, dst=dst, results=results, workspace=workspace)
    wsz = lambda m, d: hip.fse_workspace(m, 12 if d else 11, d)
else:
    decomp = lambda c, r, dst=None, results=None, workspace=None: hip.huf_decompress_batch(c, r, 32768, dst=dst, results=results, workspace=workspace)
    wsz = lambda m, d: hip.huf_workspace(m, d)
cdst, cres = comp(src)
out, dres = decomp(cdst, cres)
torch.cuda.synchronize()
assert torch.equal(out, src)

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

def run_pieces(K, NS, direction):
    bounds = [n * k // K for k in range(K + 1)]
    streams = [torch.cuda.Stream() for _ in range(NS)]
    ws = [wsz(bounds[k + 1] - bounds[k], direction == "d") for k in range(K)]
    cd2 = torch.empty_like(cdst); cr2 = torch.empty_like(cres); o2 = torch.empty_like(src); dr2 = torch.empty_like(dres)
    def go():
        main = torch.cuda.current_stream()
        e0 = torch.cuda.Event(); e0.record(main)
        for k in range(K):
            a, b = bounds[k], bounds[k + 1]
            st = streams[k % NS]
            st.wait_event(e0)
            with torch.cuda.stream(st):
                if direction == "c": comp(src[a:b], dst=cd2[a:b], results=cr2[a:b], workspace=ws[k])
                else: decomp(cdst[a:b], cres[a:b], dst=o2[a:b], results=dr2[a:b], workspace=ws[k])
        for st in streams:
            e = torch.cuda.Event(); e.record(st); main.wait_event(e)
    t = timed(go)
    if direction == "c": ok = torch.equal(cr2, cres)
    else: ok = torch.equal(o2, src)
    return t, ok

tc = timed(lambda: comp(src, dst=cdst, results=cres)); td = timed(lambda: decomp(cdst, cres, dst=out, results=dres))
print("%s P%02d single call: compress %.3f ms  decompress %.3f ms" % (codec, P, tc, td))
for K, NS in ((2, 2), (4, 2), (4, 4), (8, 2), (8, 4), (16, 4)):
    c, okc = run_pieces(K, NS, "c"); d, okd = run_pieces(K, NS, "d")
    print("  K=%2d pieces on %d streams: compress %.3f ms (%s)  decompress %.3f ms (%s)" % (K, NS, c, okc, d, okd))
