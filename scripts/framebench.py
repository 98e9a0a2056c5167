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
cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "fse_cli")
if os.path.exists(cli):
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
        data[: 64 << 20].tofile(os.path.join(d, "in"))
        t0 = time.perf_counter(); subprocess.run([cli, "-fqq", os.path.join(d, "in"), os.path.join(d, "out")], check=True, capture_output=True); t1 = time.perf_counter()
        subprocess.run([cli, "-dfqq", os.path.join(d, "out"), os.path.join(d, "back")], check=True, capture_output=True); t2 = time.perf_counter()
        print("reference tool (1 thread, files in RAM): compress %.2f GB/s, decompress %.2f GB/s" % ((64 << 20) / (t1 - t0) / 1e9, (64 << 20) / (t2 - t1) / 1e9))
