"""BASELINE configs[4] (the mixed {P02, P14, P80} corpus, FSE + Huff0 on every block) against the compiled reference, and the N > 1
path of bench.py itself (two ranks on the one GPU of the test box, gloo through host memory standing in for RCCL)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MIX = (2, 14, 80)


def test_cfg5_mixed_shard_vs_reference(hip, checker):
    """One GPU's shard of config 5 at full size (125k x 32 KB, block g drawn from P[g mod 3] with seed g + 1): both codecs' sizes and
    bytes against the compiled reference on a 1,024-block strided sample across the whole shard, every block round-trips, FSE decoded
    at the reference's default limit (maxLog 12); the shard starts at an odd global block so that the mix is not phase-aligned."""
    n, first = 125000, 125000 * 3 + 1
    src = hip.probagen_mixed(MIX, n, 32768, first_block=first)
    # the generator rule itself: three blocks of the shard against the CPU generator
    for row in (0, 1, 2, n - 1):
        g = first + row
        assert (src[row].cpu().numpy() == checker.probagen_batch(MIX[g % 3], 1, 32768, g + 1)[0]).all(), row
    idx = torch.arange(0, n, n // 1024, device=src.device)[:1024]
    host = src[idx].cpu().numpy()
    for codec in (0, 1):
        if codec == 0:
            dst, res = hip.fse_compress_batch(src, table_log=11)
            out, dres = hip.fse_decompress_batch(dst, res, 32768, max_log=12)
        else:
            dst, res = hip.huf_compress_batch(src, table_log=11)
            out, dres = hip.huf_decompress_batch(dst, res, 32768)
        assert int((res > 1).sum()) == n and int((res < 32768).sum()) == n
        assert int((dres == 32768).sum()) == n
        assert torch.equal(out, src)
        _, ores, odst = checker.compress_batch(codec, host, table_log=11)
        rh, dh = res[idx].cpu().numpy(), dst[idx].cpu().numpy()
        assert (rh == ores.astype(np.int64)).all(), codec
        for k in range(len(rh)):
            assert (dh[k][:rh[k]] == odst[k][:rh[k]]).all(), (codec, int(idx[k]))
        del dst, res, out, dres


def _run_bench(extra, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    # the driver parses the LAST stdout line out of an 8 KB tail (round 5's 21 KB line was lost): it must stay short and be the last one
    assert p.stdout.rstrip("\n").splitlines()[-1] == lines[0]
    assert len(lines[0]) < 4096, len(lines[0])
    short = json.loads(lines[0])
    with open(os.path.join(ROOT, short["detail"])) as f:
        detail = json.load(f)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype"):
        assert short[k] == detail[k], k
    return short, detail


def test_bench_two_ranks_start_from_gpus_flag():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself, reports n_gpus == 2, codes the fixed corpus of
    config 5 as two shards (strong scaling) and round-trips the with-comm variant (scatter, both codecs, gather)."""
    short, line = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--blocks", "4096", "--cfg5-blocks", "3000", "--cfg5-total", "8001",
                       "--configs", "cfg5_mixed_shard,cfg5_mixed_1M", "--no-cpu-baseline", "--no-host-inclusive", "--comm-passes", "2",
                       "--parity-blocks", "512"],
                      env={"FSEHIP_BENCH_BACKEND": "gloo"})
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    # the N > 1 line describes itself: transport, ranks that answered, config 5 compute-only and with the scatter / gather
    assert short["rccl"] == {"backend": "gloo", "world": 2, "ranks_seen": 2}
    assert short["compute_only"] == {"value": line["configs"]["cfg5_mixed_1M"]["value"], "ms_per_step": line["configs"]["cfg5_mixed_1M"]["ms_per_step"]}
    swc = short["with_comm"]
    assert swc["roundtrip_ok"] is True and swc["serial_ms"] > 0 and swc["pipelined_ms"] > 0 and swc["scatter_GBps"] > 0 and swc["gather_GBps"] > 0
    assert set(short["configs"]) == {"cfg5_mixed_shard", "cfg5_mixed_1M"} and short["configs"]["cfg5_mixed_1M"]["value"] == short["compute_only"]["value"]
    weak, strong = line["configs"]["cfg5_mixed_shard"], line["configs"]["cfg5_mixed_1M"]
    assert weak["scaling"] == "weak" and weak["blocks_per_gpu"] == 3000
    assert strong["scaling"] == "strong" and strong["corpus_blocks"] == 8001 and strong["blocks_per_gpu"] == 4001
    assert "reference-bytes" in strong["parity"] or "port-bytes" in strong["parity"]
    wc = strong["with_comm"]
    assert wc["roundtrip_ok"] is True and wc["passes"] == 2 and wc["value"] > 0
    assert set(wc["phase_ms"]) == {"1_scatter", "2_codecs", "3_gather"}
    # the pipelined variant ships packed records: what comes back over the link is the peers' payload plus 8 bytes per record offset
    pp = wc["pipelined_packed"]
    assert pp.get("error") is None and pp["roundtrip_ok"] is True and pp["passes"] == 2 and pp["value"] > 0
    n_peer = 8001 - 4001
    assert pp["scatter_bytes"] == n_peer * 32768
    assert pp["payload_bytes"] < 2 * 8001 * 32768 * 0.75 and pp["gather_bytes"] < pp["payload_bytes"]
    assert pp["gather_bytes"] - 8 * 2 * (n_peer + pp["pieces_per_shard"]) > 0 and pp["gather_bytes"] < 0.7 * pp["fixed_stride_gather_bytes"]


def test_bench_two_ranks_comm_watchdog_keeps_the_line():
    """the with-comm leg runs last, under a watchdog: when it does not finish in time (here: at once) rank 0 still prints the line --
    every other figure in place, an error record where the leg's would be -- and every rank leaves"""
    short, line = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--blocks", "2048", "--cfg5-total", "6001",
                       "--configs", "cfg5_mixed_1M", "--no-cpu-baseline", "--no-host-inclusive", "--comm-passes", "2", "--parity-blocks", "256",
                       "--comm-timeout", "0"],
                      env={"FSEHIP_BENCH_BACKEND": "gloo"})
    assert line["n_gpus"] == 2 and line["value"] > 0
    strong = line["configs"]["cfg5_mixed_1M"]
    assert strong["value"] > 0 and strong["corpus_blocks"] == 6001
    assert strong["with_comm"]["value"] is None and "did not finish within 0 s" in strong["with_comm"]["error"]
    assert short["with_comm"]["serial_ms"] is None and "did not finish" in short["with_comm"]["error"] and short["compute_only"]["value"] == strong["value"]


def test_bench_one_gpu_line_has_the_contract_fields():
    short, line = _run_bench(["--steps", "2", "--warmup", "1", "--blocks", "8192", "--configs", "cfg5_mixed_1M", "--cfg5-total", "9000",
                              "--cpu-sample-blocks", "1024", "--cpu-seconds", "0.2", "--parity-blocks", "256"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
              "roofline_encode", "cpu_baseline", "configs", "detail"):
        assert k in short, k
    for roof, side in ((short["roofline"], "decode"), (short["roofline_encode"], "encode")):
        assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0 and side in roof["kernel"]
        assert 0 < roof["frac"] < 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4 and roof["avg_launch_ms"] > 0
        assert 32768 < roof["algorithmic_bytes_per_block"] < 2 * 32768
    cb = short["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("reference", "port") and "blocks" in cb["sample"]
    assert short["config"]["blocks_per_gpu"] == 8192 and "workload" in short["config"] and "model" not in short["config"]
    c5 = short["configs"]["cfg5_mixed_1M"]
    assert c5["value"] == line["configs"]["cfg5_mixed_1M"]["value"] and 0 < c5["frac"] < 1
    assert "rccl" not in short
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["dtype"] == "u8"
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and 0 < roof["frac"] < 1 and roof["kernel"].startswith("k_fse")
    sec = roof["secondary"]
    assert sec["resident_blocks_per_cu"] >= 16 and 100 < sec["cycles_per_iteration"] < 2000 and 0 < sec["frac"] <= 1.2
    hi = line["host_inclusive"]
    assert hi["roundtrip_ok"] is True and hi["encode_GBps"] > 0 and hi["decode_GBps"] > 0
    pk = hi["packed"]                              # the same pipeline with packed results: fewer bytes over the link, same bytes back
    assert pk.get("error") is None and pk["roundtrip_ok"] is True and pk["pcie_bytes"]["encode"] < hi["pcie_bytes"]["encode"]
    assert line["configs"]["cfg5_mixed_1M"]["corpus_blocks"] == 9000 and line["configs"]["cfg5_mixed_1M"]["scaling"] == "strong"
