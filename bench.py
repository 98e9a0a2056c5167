#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X block entropy codec (contract: see task statement).

A "step" = one pass of the hot path over one batch of synthetic probagen blocks resident in HBM:
encode (FSE_compress2 / HUF_compress2 semantics: histogram, normalisation / tree, header, table, payload)
followed by decode (FSE_decompress / HUF_decompress: the reference's one-shot calls with their default
limits, i.e. maxLog 12) of every block.

Headline (`value`, every N): BASELINE.json configs[1] -- "probagen Proba14, 100k x 32KB blocks, FSE
encode+decode on 1xMI355X, bit-exact check" -- per rank (weak scaling, no collective on the data path:
every block is independent, programs/bench.c:353-364).  The other configurations ride in `configs` on
the same JSON line:
    cfg3_p80_fse        configs[2]  Proba80, FSE
    cfg4_p14_huf        configs[3]  Proba14, Huff0 4-stream
    cfg5_mixed_shard    configs[4]  mixed {P02,P14,P80} (block g: P[g mod 3], seed g+1), FSE + Huff0 on every block,
                                    125k blocks per rank (1M on 8 GPUs); for N>1 both the compute-only time (every rank
                                    generates its shard) and the with-comm time (rank 0 holds the corpus: RCCL scatter,
                                    code, RCCL gather) are reported
    fse_tl12 / huf_tl12 the tableLog `fse -b` asks for (programs/bench.c:113)
    fse_maxlog11        the headline decoded with the FSE_decompress_wksp(maxLog = 11) hint

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `value` = uncompressed MiB that went through encode AND decode per second
(whole job, all ranks), inputs resident in HBM.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

BLOCK = 32768
KERNEL_NAMES = ["k_hist", "k_fse_cprep", "k_fse_encode", "k_fse_dprep", "k_fse_decode",
                "k_huf_cprep", "k_huf_encode", "k_huf_dprep", "k_huf_decode", "k_fse_encode_wave"]
HOT = {"fse": ("k_fse_encode", "k_fse_encode_wave", "k_fse_decode"), "huf": ("k_huf_encode", "k_huf_decode")}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MIX = (2, 14, 80)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=100000, help="32 KB blocks per GPU (headline)")
    ap.add_argument("--proba", type=int, default=14)
    ap.add_argument("--workload", choices=["single", "mixed"], default="single", help="headline corpus: one distribution (--proba) or the config-5 mix")
    ap.add_argument("--codec", choices=["fse", "huf", "both"], default="fse")
    ap.add_argument("--table-log", type=int, default=11)
    ap.add_argument("--max-log", type=int, default=12, help="FSE decode limit: 12 = FSE_decompress (lib/fse_decompress.c:279-283)")
    ap.add_argument("--no-configs", action="store_true", help="headline only (skip configs 3/4/5 and the tableLog-12 variants)")
    ap.add_argument("--config-steps", type=int, default=3)
    ap.add_argument("--cfg5-blocks", type=int, default=125000, help="blocks per GPU of config 5 (1M / 8)")
    ap.add_argument("--u16-blocks", type=int, default=25000, help="blocks per GPU of the 16-bit-symbol configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-blocks", type=int, default=32768)
    ap.add_argument("--cpu-seconds", type=float, default=1.0, help="minimum timed seconds per direction and repetition")
    return ap.parse_args()


def host_threads():
    """threads the CPU baseline may use: the affinity mask, capped by the cgroup cpu quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(args, sample, codec_name):
    """The reference (oracle/_ref, kind 'reference') or our port (oracle/liboracle.so, kind 'port') on this box's host
    cores: the same blocks the GPU just coded (a bounded sample), one block per call like programs/bench.c, OpenMP over
    blocks with NUMA-local first touch, a warmed pool, >= cpu-seconds per direction, best of 3 (oracle/cpu_bench.h)."""
    from oracle.oracle import Oracle, Ref
    lib = Ref() if Ref.available() else Oracle()
    codec = 0 if codec_name == "fse" else 1
    cores = host_threads()
    n = sample.shape[0]
    best = None
    for dyn in (False, True):
        r = lib.bench_roundtrip(codec, sample, table_log=args.table_log, nthreads=cores, dynamic=dyn, min_seconds=args.cpu_seconds, reps=3)
        r["schedule"] = "dynamic,64" if dyn else "static,16"
        if best is None or r["enc_s"] + r["dec_s"] < best["enc_s"] + best["dec_s"]:
            best = r
    one = lib.bench_roundtrip(codec, sample[:max(n // 16, 64)], table_log=args.table_log, nthreads=1, min_seconds=min(args.cpu_seconds, 1.0), reps=1)
    n1 = max(n // 16, 64)
    mib, mib1 = n * BLOCK / 2.0 ** 20, n1 * BLOCK / 2.0 ** 20
    st = mib1 / (one["enc_s"] + one["dec_s"])
    value = mib / (best["enc_s"] + best["dec_s"])
    stream = lib.stream_bandwidth(1 << 30, cores, 5)
    return {
        "value": round(value, 1), "unit": "MiB/s (encode+decode round trip, uncompressed bytes)",
        "cores": best["threads"], "kind": lib.kind,
        "sample": "%d probagen P%02d blocks of 32 KB (the first blocks of the GPU workload), %s_compress2 + %s_decompress per block, OpenMP %s over "
                  "blocks, first touch in the parallel region, pool warmed on the full sample, >= %.1f s per direction, best of 3"
                  % (n, args.proba, codec_name.upper(), codec_name.upper(), best["schedule"], args.cpu_seconds),
        "encode_MiBps": round(mib / best["enc_s"], 1), "decode_MiBps": round(mib / best["dec_s"], 1),
        "single_thread_encode_MiBps": round(mib1 / one["enc_s"], 1), "single_thread_decode_MiBps": round(mib1 / one["dec_s"], 1),
        "single_thread_value": round(st, 1),
        "scaling_efficiency": round(value / (best["threads"] * st), 3),
        "host_copy_bandwidth_GBps": round(stream, 1),
        "note": "efficiency = all-core value / (threads x single-thread value); threads = affinity mask capped by the cgroup quota "
                "(SMT siblings count as threads); the round trip moves ~2.5 bytes of memory traffic per uncompressed byte",
    }


class Codec:
    """encode / decode closures of one codec over preallocated device buffers (views of the shared pools)"""

    def __init__(self, hip, name, src, pools, table_log, max_log):
        from finitestateentropy_amd.api import fse_compress_bound, huf_compress_bound
        self.hip, self.name, self.src, self.tl, self.max_log = hip, name, src, table_log, max_log
        nb = src.shape[0]
        self.cap = fse_compress_bound(BLOCK) if name == "fse" else huf_compress_bound(BLOCK)
        self.dst = pools["dst_" + name][:nb * self.cap].view(nb, self.cap)
        self.res = pools["res_" + name][:nb]
        self.out = pools["out_" + name][:nb * BLOCK].view(nb, BLOCK)
        self.dres = pools["dres_" + name][:nb]
        dev = src.device
        if name == "fse":
            self.ws_c = hip.fse_workspace(nb, table_log, False, dev)
            self.ws_d = hip.fse_workspace(nb, max_log, True, dev)
        else:
            self.ws_c = hip.huf_workspace(nb, False, dev)
            self.ws_d = hip.huf_workspace(nb, True, dev)

    def encode(self):
        if self.name == "fse":
            self.hip.fse_compress_batch(self.src, self.tl, dst=self.dst, results=self.res, workspace=self.ws_c)
        else:
            self.hip.huf_compress_batch(self.src, self.tl, dst=self.dst, results=self.res, workspace=self.ws_c)

    def decode(self):
        if self.name == "fse":
            self.hip.fse_decompress_batch(self.dst, self.res, BLOCK, max_log=self.max_log, dst=self.out, results=self.dres, workspace=self.ws_d)
        else:
            self.hip.huf_decompress_batch(self.dst, self.res, BLOCK, dst=self.out, results=self.dres, workspace=self.ws_d)


def check_parity(cd, rank, n_check=1024):
    """untimed gates: round trip on every block; encoder bytes and return values against the compiled reference (or the
    oracle when oracle/_ref is absent) on a strided sample across the whole batch"""
    nb = cd.src.shape[0]
    assert bool((cd.dres == BLOCK).all()), "%s decode return values wrong" % cd.name
    assert torch.equal(cd.out, cd.src), "%s decode(encode(x)) != x" % cd.name
    parity = "roundtrip-all-blocks"
    if rank != 0:
        return parity
    try:
        from oracle.oracle import Oracle, Ref
        lib = Ref() if Ref.available() else Oracle()
    except OSError:
        return parity + "(checker unavailable)"
    idx = torch.arange(0, nb, max(1, nb // n_check), device=cd.src.device)[:n_check]
    host = cd.src[idx].cpu().numpy()
    _, ores, odst = lib.compress_batch(0 if cd.name == "fse" else 1, host, table_log=cd.tl)
    rh, dh = cd.res[idx].cpu().numpy(), cd.dst[idx].cpu().numpy()
    assert (rh == ores.astype(np.int64)).all(), "%s encode sizes differ from the CPU %s" % (cd.name, lib.kind)
    for b in range(len(rh)):
        assert (dh[b][:rh[b]] == odst[b][:rh[b]]).all(), "%s encode bytes differ from the CPU %s (block %d)" % (cd.name, lib.kind, int(idx[b]))
    return parity + "+%s-bytes-%d-blocks-strided" % (lib.kind, len(rh))


def u16_case(hip, dev, n_blocks, steps, barrier, reduce_max, world, rank):
    """the 16-bit-symbol coder (lib/fseU16.c; what programs/bench.c:221,248 times in its U16 mode) -- round trip of
    n_blocks x 16384 symbols (= 32 KB) per GPU.  Corpus: 256 distinct blocks from the generator of programs/fuzzerU16.c:107-134
    (p = 0.08 from symbol 240, wrapping inside the 287-symbol alphabet), tiled; bytes checked against the compiled reference."""
    nsym, distinct = BLOCK // 2, 256
    rng = np.random.default_rng(16)
    table = np.zeros(4096, np.uint16)
    remaining, pos, val = 4096, 0, 240
    while remaining:
        k = int(remaining * 0.08) + 1
        table[pos:pos + k] = val
        pos += k; remaining -= k
        val = val + 1 if val + 1 < 286 else 1
    host = table[rng.integers(0, 4096, (distinct, nsym))]
    base = torch.from_numpy(host.view(np.int16)).to(dev)
    src = base.repeat((n_blocks + distinct - 1) // distinct, 1)[:n_blocks].contiguous()
    cdst, cres = hip.fse_compress_u16_batch(src)
    out, dres = hip.fse_decompress_u16_batch(cdst, cres, nsym)
    torch.cuda.synchronize()
    assert bool((dres == nsym).all()) and torch.equal(out, src), "u16 decode(encode(x)) != x"
    parity = "roundtrip-all-blocks"
    if rank == 0:
        try:
            from oracle.oracle import Ref
            if Ref.available():
                ref = Ref()
                ch, rh = cdst[:distinct:4].cpu().numpy(), cres[:distinct:4].cpu().numpy()
                for b in range(len(rh)):
                    rr, rout = ref.fse_compress_u16(host[4 * b], 0, 0)
                    assert rr == int(rh[b]) and (rout[:rr] == ch[b][:rr]).all(), "u16 encode differs from the reference (block %d)" % (4 * b)
                parity += "+reference-bytes-%d-blocks" % len(rh)
        except OSError:
            parity += "(checker unavailable)"
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        hip.fse_compress_u16_batch(src, dst=cdst, results=cres); ev[2 * i + 1].record()
        hip.fse_decompress_u16_batch(cdst, cres, nsym, dst=out, results=dres); ev[2 * i + 2].record()
    barrier()
    elapsed = time.perf_counter() - t0
    enc = sum(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(steps)) * 1e-3
    dec = sum(ev[2 * i + 1].elapsed_time(ev[2 * i + 2]) for i in range(steps)) * 1e-3
    elapsed, enc, dec = reduce_max([elapsed, enc, dec])
    total = world * n_blocks * BLOCK * steps
    return {"value": round(total / 2.0 ** 20 / elapsed, 1), "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps, "blocks_per_gpu": n_blocks,
            "encode_GBps": round(total / enc / 1e9, 2), "decode_GBps": round(total / dec / 1e9, 2),
            "compressed_bytes_per_block": round(float(cres.sum().item()) / n_blocks, 1), "parity": parity,
            "workload": "16-bit symbols (lib/fseU16.c): %d x 16384 symbols per GPU, 287-symbol alphabet (fuzzerU16's generator, p = 0.08), "
                        "FSE_compressU16 + FSE_decompressU16, default table log 12; one tANS state per block: the encoder splits the chain across a wave, the decoder runs one lane per block" % n_blocks}


def run_case(hip, codecs, steps, warmup, barrier, rank, check=True):
    """time `steps` steps (each: encode + decode of every codec in `codecs`), bracketed by barrier + synchronize.
    Returns the raw timings of this rank and the per-kernel probe."""
    for _ in range(warmup):
        for cd in codecs:
            cd.encode(); cd.decode()
    barrier()
    hip.lib.FSEHIP_probe_begin()
    nev = 2 * len(codecs) * steps + 1
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(nev)]
    t0 = time.perf_counter()
    ev[0].record()
    k = 1
    for _ in range(steps):
        for cd in codecs:
            cd.encode(); ev[k].record(); k += 1
            cd.decode(); ev[k].record(); k += 1
    barrier()
    elapsed = time.perf_counter() - t0
    ms = (C.c_double * 16)(); launches = (C.c_uint * 16)()
    hip.lib.FSEHIP_probe_collect(ms, launches)
    enc_s = {cd.name: 0.0 for cd in codecs}; dec_s = {cd.name: 0.0 for cd in codecs}
    k = 1
    for _ in range(steps):
        for cd in codecs:
            enc_s[cd.name] += ev[k - 1].elapsed_time(ev[k]) / 1e3; k += 1
            dec_s[cd.name] += ev[k - 1].elapsed_time(ev[k]) / 1e3; k += 1
    per = {KERNEL_NAMES[i]: (ms[i], launches[i]) for i in range(len(KERNEL_NAMES)) if launches[i]}
    out = {"elapsed": elapsed, "enc_s": enc_s, "dec_s": dec_s, "per": per, "parity": {}, "csize": {}}
    for cd in codecs:
        out["csize"][cd.name] = float(cd.res.sum().item()) / cd.src.shape[0]
        if check:
            out["parity"][cd.name] = check_parity(cd, rank)
    return out


def roofline(per, codec_name, mean_csize, nb, steps, traffic_tag=None):
    hot = [k for k in HOT[codec_name] if k in per]
    dom = max(hot, key=lambda k: per[k][0])
    dom_ms, dom_launches = per[dom]
    alg = BLOCK + mean_csize                                      # read input once + write output once
    blocks_per_launch = nb * steps / dom_launches
    achieved = alg * blocks_per_launch / (dom_ms / dom_launches * 1e-3) / 1e9
    traffic = None
    by_size = None
    if traffic_tag is not None:
        tpath = os.path.join(ROOT, "profiles", "traffic_%s%s.json" % (dom, traffic_tag))
        if os.path.exists(tpath):
            try:   # PMC counters come from a separate rocprofv3 --pmc pass (scripts/pmc_summary.py -> profiles/); per launch like `achieved`
                rec = json.load(open(tpath))
                traffic = round(rec.get("hbm_bytes_per_block") * blocks_per_launch)
                by_size = rec.get("request_size_bytes_per_block")
            except Exception:
                traffic = None
    out = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "algorithmic_bytes_per_block": round(alg, 1),
           "blocks_per_launch": round(blocks_per_launch, 1), "avg_launch_ms": round(dom_ms / dom_launches, 4)}
    if dom.endswith("decode"):    # the decoders run one launch per class of blocks; HIP events bracket the group
        out["note"] = ("avg_launch_ms brackets the kernel's launches of one step (one per decoder class: launches over classes without blocks "
                       "return in microseconds and show up as extra calls in rocprofv3's table; profiles/*_pmc.md lists the per-pass totals)")
    if by_size:   # cross-check of `traffic`: the L2's memory-side requests counted by request size (exact bytes, profiles/*_pmc.md)
        out["traffic_by_request_size"] = {"read_bytes_per_block": by_size["read"], "write_bytes_per_block": by_size["write"]}
    return out


def summarize(r, codecs, nb, steps, world, reduce_max):
    """per-configuration record from the raw timings (max over ranks)"""
    names = [cd.name for cd in codecs]
    vals = [r["elapsed"]] + [r["enc_s"][n] for n in names] + [r["dec_s"][n] for n in names]
    vals = reduce_max(vals)
    elapsed = vals[0]
    total_bytes = world * nb * BLOCK * steps
    rec = {"value": round(total_bytes / 2.0 ** 20 / elapsed, 1), "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps, "blocks_per_gpu": nb}
    for i, n in enumerate(names):
        rec["%s_encode_GBps" % n] = round(total_bytes / vals[1 + i] / 1e9, 2)
        rec["%s_decode_GBps" % n] = round(total_bytes / vals[1 + len(names) + i] / 1e9, 2)
        rec["%s_compressed_bytes_per_block" % n] = round(r["csize"][n], 1)
    rec["kernel_ms_per_step"] = {k: round(v[0] / steps, 3) for k, v in r["per"].items()}
    dom_codec = max(names, key=lambda n: max(r["per"].get(k, (0, 0))[0] for k in HOT[n]))
    rec["roofline"] = roofline(r["per"], dom_codec, r["csize"][dom_codec], nb, steps)
    if r["parity"]:
        rec["parity"] = "; ".join("%s: %s" % (n, r["parity"][n]) for n in names)
    return rec, vals


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # (FSEHIP_BENCH_BACKEND=gloo lets several ranks share one GPU for a smoke test of the N>1 path; the driver's runs use RCCL)
    backend = os.environ.get("FSEHIP_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    from finitestateentropy_amd import shard
    from finitestateentropy_amd.api import FseHip, fse_compress_bound, huf_compress_bound
    hip = FseHip()
    dev = torch.device("cuda", local_rank)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(values):
        return shard.max_over_ranks(values, dev, world)

    nb = args.blocks
    nmax = nb if args.no_configs else max(nb, args.cfg5_blocks)
    # device pools shared by every configuration (views are taken per case)
    pools = {}
    want = ("fse", "huf") if (args.codec == "both" or not args.no_configs) else (args.codec,)
    for n in want:
        cap = fse_compress_bound(BLOCK) if n == "fse" else huf_compress_bound(BLOCK)
        pools["dst_" + n] = torch.empty(nmax * cap, dtype=torch.uint8, device=dev)
        pools["res_" + n] = torch.empty(nmax, dtype=torch.int64, device=dev)
        pools["out_" + n] = torch.empty(nmax * BLOCK, dtype=torch.uint8, device=dev)
        pools["dres_" + n] = torch.empty(nmax, dtype=torch.int64, device=dev)
    srcpool = torch.empty(nmax * BLOCK, dtype=torch.uint8, device=dev)

    def gen(proba, n, first_block):
        """rank-local shard of the conceptual global corpus: global block g = first_block + row, seed g + 1"""
        view = srcpool[:n * BLOCK].view(n, BLOCK)
        if proba == "mixed":
            view.copy_(hip.probagen_mixed(MIX, n, BLOCK, first_block=first_block, device=dev))
        else:
            hip.probagen_batch(proba, n, BLOCK, first_seed=1 + first_block, out=view)
        return view

    # ---------------- headline: configs[1] per rank
    head_proba = "mixed" if args.workload == "mixed" else args.proba
    src = gen(head_proba, nb, rank * nb)
    head_names = ("fse", "huf") if args.codec == "both" else (args.codec,)
    codecs = [Codec(hip, n, src, pools, args.table_log, args.max_log) for n in head_names]
    r = run_case(hip, codecs, args.steps, args.warmup, barrier, rank)
    head, vals = summarize(r, codecs, nb, args.steps, world, reduce_max)
    elapsed = vals[0]
    dom_codec = max(head_names, key=lambda n: max(r["per"].get(k, (0, 0))[0] for k in HOT[n]))
    head_roof = roofline(r["per"], dom_codec, r["csize"][dom_codec], nb, args.steps, traffic_tag="")
    cpu_sample = None
    if world == 1 and not args.no_cpu_baseline and head_proba != "mixed":
        cpu_sample = src[:min(args.cpu_sample_blocks, nb)].cpu().numpy()
    del codecs

    configs = {}
    if not args.no_configs:
        cs = args.config_steps

        def case(key, proba, names, n, table_log=args.table_log, max_log=args.max_log, desc=""):
            s = gen(proba, n, rank * n)
            cds = [Codec(hip, nm, s, pools, table_log, max_log) for nm in names]
            rr = run_case(hip, cds, cs, 1, barrier, rank)
            rec, _ = summarize(rr, cds, n, cs, world, reduce_max)
            rec["workload"] = desc
            configs[key] = rec
            return s, cds

        case("fse_maxlog11", args.proba, ("fse",), nb, max_log=max(args.table_log, 9),
             desc="headline workload decoded with FSE_decompress_wksp(maxLog = 11) (caller promises tableLog <= 11)")
        case("cfg3_p80_fse", 80, ("fse",), nb, desc="BASELINE configs[2]: probagen Proba80, %d x 32KB blocks per GPU, FSE encode+decode" % nb)
        case("cfg4_p14_huf", 14, ("huf",), nb, desc="BASELINE configs[3]: probagen Proba14, %d x 32KB blocks per GPU, Huff0 4-stream encode + HUF_decompress" % nb)
        case("fse_tl12", 14, ("fse",), nb, table_log=12, desc="Proba14, FSE with tableLog 12 (what `fse -b` requests, programs/bench.c:113)")
        case("huf_tl12", 2, ("huf",), nb, table_log=12, desc="Proba02 (256 symbols), Huff0 with tableLog 12 (HUF_TABLELOG_MAX)")
        n5 = args.cfg5_blocks
        s5, cds5 = case("cfg5_mixed_shard", "mixed", ("fse", "huf"), n5,
                        desc="BASELINE configs[4]: probagen mixed {P02,P14,P80} (block g: P[g mod 3], seed g+1), %d x 32KB blocks per GPU "
                             "(%d in all), FSE and Huff0 round trip of every block, sharded by contiguous block ranges; compute-only "
                             "(each rank generates its shard, no collective)" % (n5, n5 * world))
        configs["cfg5_mixed_shard"]["value_note"] = "value counts each block once per step although it goes through both codecs"
        if world > 1:
            # with-comm variant: rank 0 holds the whole corpus; RCCL scatter of the raw blocks, FSE + Huff0 round trip on every rank,
            # RCCL gather of the compressed slots and sizes of both codecs (finitestateentropy_amd/shard.py, star over xGMI)
            n_total = n5 * world
            corpus = hip.probagen_mixed(MIX, n_total, BLOCK, first_block=0, device=dev) if rank == 0 else None
            barrier()
            t0 = time.perf_counter()
            mine, gathered = shard.sharded_codec_job(corpus, n_total, BLOCK, rank, world, dev, cds5)
            barrier()
            t_comm = reduce_max([time.perf_counter() - t0])[0]
            ok = shard.sharded_job_ok(mine, gathered, cds5, n_total, BLOCK, rank, world)
            configs["cfg5_mixed_shard"]["with_comm"] = {
                "value": round(n_total * BLOCK / 2.0 ** 20 / t_comm, 1), "ms": round(t_comm * 1e3, 2), "roundtrip_ok": bool(ok),
                "what": "rank 0 holds %d blocks: scatter (point-to-point per peer) + FSE and Huff0 encode+decode + gather of both "
                        "codecs' fixed-stride compressed slots and sizes; one pass, untimed warm-up = the compute-only run above" % n_total}
            del corpus, gathered
        del s5, cds5
        if hasattr(hip, "fse_compress_u16_batch"):
            configs["fse_u16"] = u16_case(hip, dev, args.u16_blocks, cs, barrier, reduce_max, world, rank)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    name = {"fse": "FSE", "huf": "Huff0 4-stream", "both": "FSE and Huff0"}[args.codec]
    line = {
        "metric": "MB/s encode+decode per GPU on 32KB probagen blocks; bit-exact vs CPU ref",
        "value": head["value"], "unit": "MiB/s of uncompressed data through encode+decode (whole job)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "probagen %s, %d x 32KB blocks per GPU, %s encode+decode (one-shot calls at the reference's default limits: "
                               "tableLog %d, decode maxLog %d), bit-exact check"
                               % ("mixed P02/P14/P80" if head_proba == "mixed" else "Proba%02d" % args.proba, nb, name, args.table_log, args.max_log),
                   "blocks_per_gpu": nb, "block_bytes": BLOCK, "codec": args.codec, "parity": head.get("parity"),
                   "compressed_bytes_per_block": head["%s_compressed_bytes_per_block" % head_names[0]]},
        "encode_GBps": head["%s_encode_GBps" % head_names[0]], "decode_GBps": head["%s_decode_GBps" % head_names[0]],
        "roofline": head_roof,
        "kernel_ms_per_step": head["kernel_ms_per_step"],
    }
    if len(head_names) > 1:
        for n in head_names[1:]:
            line["%s_encode_GBps" % n] = head["%s_encode_GBps" % n]; line["%s_decode_GBps" % n] = head["%s_decode_GBps" % n]
    if configs:
        line["configs"] = configs
    if cpu_sample is not None:
        try:
            line["cpu_baseline"] = cpu_baseline(args, cpu_sample, head_names[0])
        except Exception as e:  # the baseline is a report, never a reason to lose the GPU line
            line["cpu_baseline"] = {"value": None, "unit": "MiB/s", "cores": host_threads(), "kind": "unavailable", "sample": repr(e)}
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
