"""development aid: host-buffer .fse frame throughput (PCIe + host assembly inclusive) next to the reference tool"""
import sys, os, time, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finitestateentropy_amd.api import FseHip
from oracle.oracle import Oracle
hip = FseHip(); orc = Oracle()
n_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
data = orc.probagen_batch(14, n_blocks, 32768, 1).reshape(-1)
for codec, name in ((0, "fse"), (1, "huf")):
    hip.frame_compress(data[:1 << 20], 5, codec)                       # warm-up (context, allocations)
    t0 = time.perf_counter(); r, frame = hip.frame_compress(data, 5, codec); t1 = time.perf_counter()
    r2, back = hip.frame_decompress(frame[:r], data.size); t2 = time.perf_counter()
    assert r2 == data.size and (back[:r2] == data).all()
    print("%s frame: %d MB -> %d MB; compress %.2f GB/s, decompress %.2f GB/s (host buffers, PCIe inclusive)" % (
        name, data.size >> 20, r >> 20, data.size / (t1 - t0) / 1e9, data.size / (t2 - t1) / 1e9))
# many frames per call: FSEHIP_frame_*_batch over 8 frames of an eighth of the data each (8 host threads), and over 32 small ones
import ctypes as C
def batch_rate(parts, codec, threads):
    n = len(parts)
    PA, SA = C.c_void_p * n, C.c_size_t * n
    bound = [int(hip.lib.FSEHIP_frame_compressBound(C.c_size_t(x.size), C.c_uint(5))) for x in parts]
    outs = [np.empty(b, np.uint8) for b in bound]; backs = [np.empty(x.size, np.uint8) for x in parts]
    res = SA(); res2 = SA()
    fc, fd = hip.lib.FSEHIP_frame_compress_batch, hip.lib.FSEHIP_frame_decompress_batch
    fc.restype = fd.restype = C.c_size_t
    args_c = (PA(*[o.ctypes.data for o in outs]), SA(*bound), PA(*[x.ctypes.data for x in parts]), SA(*[x.size for x in parts]), res, C.c_size_t(n), C.c_uint(5), C.c_int(codec), C.c_uint(threads))
    fc(*args_c)                                                         # warm-up: the workers' arenas and streams
    t0 = time.perf_counter(); assert fc(*args_c) == 0; t1 = time.perf_counter()
    args_d = (PA(*[b.ctypes.data for b in backs]), SA(*[x.size for x in parts]), PA(*[o.ctypes.data for o in outs]), SA(*[int(res[i]) for i in range(n)]), res2, C.c_size_t(n), C.c_uint(threads))
    fd(*args_d)
    t2 = time.perf_counter(); assert fd(*args_d) == 0; t3 = time.perf_counter()
    assert all(int(res2[i]) == parts[i].size and (backs[i] == parts[i]).all() for i in range(n))
    total = sum(x.size for x in parts)
    return total / (t1 - t0) / 1e9, total / (t3 - t2) / 1e9
for codec, name in ((0, "fse"), (1, "huf")):
    c, d = batch_rate([data], codec, 1)
    print("%s frame, destinations already touched (no first-touch page faults in the timed call): compress %.2f GB/s, decompress %.2f GB/s" % (name, c, d))
    for nparts, threads in ((8, 0), (8, 2), (8, 4), (8, 8), (32, 4), (32, 8), (8, 1)):
        parts = [np.ascontiguousarray(x) for x in np.array_split(data, nparts)]
        c, d = batch_rate(parts, codec, threads)
        print("%s frames, batch of %d x %d MB on %d host threads: compress %.2f GB/s, decompress %.2f GB/s" % (name, nparts, parts[0].size >> 20, threads, c, d))
cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "fse_cli")
if os.path.exists(cli):
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
        data[: 64 << 20].tofile(os.path.join(d, "in"))
        t0 = time.perf_counter(); subprocess.run([cli, "-fqq", os.path.join(d, "in"), os.path.join(d, "out")], check=True, capture_output=True); t1 = time.perf_counter()
        subprocess.run([cli, "-dfqq", os.path.join(d, "out"), os.path.join(d, "back")], check=True, capture_output=True); t2 = time.perf_counter()
        print("reference tool (1 thread, files in RAM): compress %.2f GB/s, decompress %.2f GB/s" % ((64 << 20) / (t1 - t0) / 1e9, (64 << 20) / (t2 - t1) / 1e9))
