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
    cfg5_mixed_1M       configs[4] AS NAMED: the fixed 1M-block mixed {P02,P14,P80} corpus (block g: P[g mod 3], seed g+1),
                                    FSE + Huff0 on every block, rank r codes shard_range(1M, r, N) -> "scaling": "strong"
                                    (N = 1: the whole corpus on one GPU).  Compute-only (every rank generates its shard) and,
                                    for N > 1, with-comm (rank 0 holds the corpus: grouped RCCL scatter, code, grouped RCCL
                                    gather; communicators warmed by one untimed pass, then >= 3 timed passes)
    cfg5_mixed_shard    the same mix at 125k blocks per rank whatever N ("scaling": "weak"; 1M blocks on 8 GPUs)
    fse_tl12 / huf_tl12 the tableLog `fse -b` asks for (programs/bench.c:113)
    fse_maxlog11        the headline decoded with the FSE_decompress_wksp(maxLog = 11) hint

    python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE in the environment: bench.py starts the
                                                            N ranks itself through torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `value` = uncompressed MiB that went through encode AND decode per second
(whole job, all ranks), inputs resident in HBM.
"""
import argparse
import datetime
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
    ap.add_argument("--configs", default="", help="comma-separated subset of the `configs` keys to run (default: all)")
    ap.add_argument("--ut-keys", default="", help="comma-separated subset of the using_tables records (fse_p14, fse_p80, huf_p14; default: all) -- scripts/profile.sh "
                                                  "collects their counters one record per run")
    ap.add_argument("--config-steps", type=int, default=0, help="steps of the `configs` entries (0 = the headline's --steps / --warmup)")
    ap.add_argument("--cfg5-blocks", type=int, default=125000, help="blocks per GPU of the weak-scaling config-5 record (1M / 8)")
    ap.add_argument("--cfg5-total", type=int, default=1000000, help="blocks of the fixed config-5 corpus (strong scaling; BASELINE configs[4])")
    ap.add_argument("--comm-passes", type=int, default=3, help="timed passes of the with-comm variant (after one untimed pass)")
    ap.add_argument("--comm-pieces", type=int, default=4, help="pieces per shard of the pipelined with-comm variant")
    ap.add_argument("--u16-blocks", type=int, default=25000, help="blocks per GPU of the 16-bit-symbol configuration")
    ap.add_argument("--parity-blocks", type=int, default=0, help="blocks whose encoder bytes are compared with the CPU reference, untimed (0 = every block of the "
                                                                 "headline and of configs 2-4; the mixed 1M-block configurations use --cfg5-parity-blocks)")
    ap.add_argument("--comm-timeout", type=int, default=240, help="seconds the with-comm leg of config 5 (N > 1) may take before the line is printed without it")
    ap.add_argument("--cfg5-parity-blocks", type=int, default=0, help="blocks per codec of the config-5 records compared with the CPU reference: 0 = ALL of them "
                                                                      "(chunked and untimed; the default), N = a strided sample of N")
    ap.add_argument("--no-host-inclusive", action="store_true", help="skip the pinned-host H2D + kernels + D2H figure")
    ap.add_argument("--plain", action="store_true", help="profiler runs (scripts/profile.sh): warm-up + timed steps only -- no event-probe pass, no instrumented "
                                                          "decode pass, no host-inclusive / CPU legs -- so that every kernel is launched exactly warmup + steps times")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-blocks", type=int, default=32768)
    ap.add_argument("--cpu-seconds", type=float, default=1.0, help="minimum timed seconds per direction and repetition")
    return ap.parse_args()


def launch_ranks_if_needed(args):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE: become `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    (one rank per GPU, rendezvous on 127.0.0.1).  Under a launcher the world it made must be the one asked for."""
    if "WORLD_SIZE" in os.environ:
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            sys.exit("bench.py: launched with WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
        return
    if args.gpus <= 1:
        return
    import socket
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, cmd)


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
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                model = ln.split(":", 1)[1].strip(); break
    except OSError:
        pass
    return {
        "cpu_model": model, "host_logical_cpus": os.cpu_count(),
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

    def piece(self, lo, hi):
        """the same codec over rows [lo, hi) of the batch (views of the same buffers, the same workspaces): shard.sharded_codec_job_pipelined"""
        import copy
        pc = copy.copy(self)
        pc.src = self.src[lo:hi] if self.src is not None else None
        pc.dst, pc.res, pc.out, pc.dres = self.dst[lo:hi], self.res[lo:hi], self.out[lo:hi], self.dres[lo:hi]
        return pc

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


def reference_bytes_check(lib, codec_name, table_log, src, dst, res, n_check=0, chunk=16384, hdr=None, hdr_sizes=None):
    """encoder bytes and return values of blocks against the compiled reference (the restatement when oracle/_ref is absent), untimed:
    all blocks (n_check = 0) or n_check blocks strided across the batch, in chunks -- the reference codes a chunk on the host cores,
    the comparison itself runs on the device.  With `hdr` / `hdr_sizes` (the using-table calls) the reference's block is header then
    payload: the device's header bytes and payload bytes are compared with the two parts.  Returns the number of blocks checked."""
    nb = src.shape[0]
    dev = src.device
    idx = torch.arange(nb, device=dev) if not n_check or n_check >= nb else torch.arange(0, nb, max(1, nb // n_check), device=dev)[:n_check]
    codec = 0 if codec_name == "fse" else 1
    cols = torch.arange(dst.shape[1], device=dev)
    for lo in range(0, idx.numel(), chunk):
        sel = idx[lo:lo + chunk]
        host = src[sel].cpu().numpy()
        _, ores, odst = lib.compress_batch(codec, host, table_log=table_log, nthreads=host_threads(), cap=dst.shape[1])
        ref_res = torch.from_numpy(ores.astype(np.int64)).to(dev)
        ref_dst = torch.from_numpy(odst).to(dev)
        coded = (ref_res > 1) & (ref_res < (1 << 40))
        if hdr is None:
            got = res[sel]
            assert torch.equal(got, ref_res), "%s encode return values differ from the CPU %s (block %d)" % (
                codec_name, lib.kind, int(sel[(got != ref_res).nonzero()[0, 0]]))
            live = cols[None, :] < (ref_res * coded)[:, None]
            bad = ((dst[sel] != ref_dst) & live).any(dim=1)
        else:
            h = hdr_sizes[sel]
            assert bool((coded == (h > 1)).all()), "%s table builder and the CPU %s disagree on which blocks are coded" % (codec_name, lib.kind)
            assert bool(((res[sel] + h == ref_res) | ~coded).all()), "%s header + payload sizes differ from the CPU %s" % (codec_name, lib.kind)
            hc = cols[None, :hdr.shape[1]]
            bad = ((hdr[sel] != ref_dst[:, :hdr.shape[1]]) & (hc < (h * coded)[:, None])).any(dim=1)
            # payload byte j of the device = byte h + j of the reference's block
            shifted = torch.gather(ref_dst, 1, (cols[None, :] + (h * coded)[:, None]).clamp(max=ref_dst.shape[1] - 1))
            bad |= ((dst[sel] != shifted) & (cols[None, :] < (res[sel] * coded)[:, None])).any(dim=1)
        assert not bool(bad.any()), "%s encode bytes differ from the CPU %s (block %d)" % (codec_name, lib.kind, int(sel[bad.nonzero()[0, 0]]))
    return int(idx.numel())


def checker_lib():
    try:
        from oracle.oracle import Oracle, Ref
        return Ref() if Ref.available() else Oracle()
    except OSError:
        return None


def check_parity(cd, rank, n_check=0):
    """untimed gates: round trip on every block; encoder bytes and return values of EVERY block (n_check = 0; else a strided sample)
    against the compiled reference (or the oracle when oracle/_ref is absent)"""
    nb = cd.src.shape[0]
    assert bool((cd.dres == BLOCK).all()), "%s decode return values wrong" % cd.name
    assert torch.equal(cd.out, cd.src), "%s decode(encode(x)) != x" % cd.name
    parity = "roundtrip-all-blocks"
    if rank != 0:
        return parity, 0
    lib = checker_lib()
    if lib is None:
        return parity + "(checker unavailable)", 0
    n = reference_bytes_check(lib, cd.name, cd.tl, cd.src, cd.dst, cd.res, n_check)
    return parity + "+%s-bytes-%s" % (lib.kind, "all-%d-blocks" % n if n == nb else "%d-blocks-strided" % n), n


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
                # every DISTINCT block against the reference's bytes (the corpus tiles them); the tiles against the first one on the device
                ch, rh = cdst[:distinct].cpu().numpy(), cres[:distinct].cpu().numpy()
                for b in range(len(rh)):
                    rr, rout = ref.fse_compress_u16(host[b], 0, 0)
                    assert rr == int(rh[b]) and (rout[:rr] == ch[b][:rr]).all(), "u16 encode differs from the reference (block %d)" % b
                nd = min(distinct, n_blocks)
                assert torch.equal(cres, cres[:nd].repeat((n_blocks + nd - 1) // nd)[:n_blocks]), "u16 encode: tiled blocks give different sizes"
                cols = torch.arange(cdst.shape[1], device=dev)[None, :]
                for lo in range(0, n_blocks, 4096):
                    hi = min(lo + 4096, n_blocks)
                    first = cdst[torch.arange(lo, hi, device=dev) % nd]
                    assert not bool(((cdst[lo:hi] != first) & (cols < cres[lo:hi, None])).any()), "u16 encode: tiled blocks give different bytes"
                parity += "+reference-bytes-all-%d-distinct-blocks(tiled-to-%d-and-compared-on-device)" % (len(rh), n_blocks)
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
    pmc = {}
    for k in ("k_u16_cprep", "k_u16_encode_wave", "k_u16_dprep", "k_u16_decode_lds"):          # the builder's PMC passes (scripts/profile.sh), a cross-reference
        tp = os.path.join(ROOT, "profiles", "traffic_%s_u16run.json" % k)
        if os.path.exists(tp):
            try:
                pmc[k] = json.load(open(tp)).get("hbm_bytes_per_block")
            except Exception:
                pass
    return {"pmc_hbm_bytes_per_block": pmc or None,
            "value": round(total / 2.0 ** 20 / elapsed, 1), "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps, "blocks_per_gpu": n_blocks,
            "encode_GBps": round(total / enc / 1e9, 2), "decode_GBps": round(total / dec / 1e9, 2),
            "compressed_bytes_per_block": round(float(cres.sum().item()) / n_blocks, 1), "parity": parity,
            "workload": "16-bit symbols (lib/fseU16.c): %d x 16384 symbols per GPU, 287-symbol alphabet (fuzzerU16's generator, p = 0.08), "
                        "FSE_compressU16 + FSE_decompressU16 at the default limits (FSE_optimalTableLog gives these blocks table log 11: default 12, capped by highbit(16383) - 2); one tANS state per block: the encoder splits the chain across a wave, the decoder runs one lane per block" % n_blocks}


def using_tables_case(hip, codec_name, proba, src, pools, table_log, steps, warmup, barrier, reduce_max, world, rank, key=""):
    """The functions north_star names, in the form it names them (lib/fse.h:174,247, lib/huf.h:190,275; what programs/fullbench.c:805-814,
    851-862,897-905,987-998 time separately): tables built ONCE on the device (FSEHIP_*_build*Table_batch: the library's own prepare
    kernels behind calls of their own), then `steps` timed passes of *_compress_usingCTable_batch + *_decompress_usingDTable_batch over
    the same blocks.  Parity: every block's header + payload against the reference's one-shot output, every block round-tripped."""
    from finitestateentropy_amd.api import fse_compress_bound, huf_compress_bound
    nb = src.shape[0]
    dev = src.device
    fse = codec_name == "fse"
    cap = fse_compress_bound(BLOCK) if fse else huf_compress_bound(BLOCK)
    dst = pools["dst_" + codec_name][:nb * cap].view(nb, cap)
    res = pools["res_" + codec_name][:nb]
    out = pools["out_" + codec_name][:nb * BLOCK].view(nb, BLOCK)
    dres = pools["dres_" + codec_name][:nb]
    if fse:
        ct, hdr, hres = hip.fse_build_ctable_batch(src, table_log=table_log)
        dt, dtres = hip.fse_build_dtable_batch(hdr, hres, max_log=table_log)
        # maxTableLog of the decode call = what the caller promises about its tables: it built them with `table_log`, so that is the bound (one launch;
        # FSEHIP_BENCH_UT_MAXLOG=12 = no promise, FSE_MAX_TABLELOG: two more launches look for tableLog-12 tables, +0.4 ms per 100k blocks)
        dec_maxlog = int(os.environ.get("FSEHIP_BENCH_UT_MAXLOG", str(table_log)))
    else:
        ct, hdr, hres = hip.huf_build_ctable_batch(src, table_log=table_log)
        dt, dtres = hip.huf_read_dtable_x1_batch(hdr, hres, max_table_log=table_log)
    torch.cuda.synchronize()
    assert bool((hres > 1).all()) and bool((dtres == hres).all()), "%s table builders refused a block of the workload" % codec_name

    def encode():
        if fse:
            hip.fse_compress_using_ctable_batch(src, ct, max_table_log=table_log, dst=dst, results=res)
        else:
            hip.huf_compress4x_using_ctable_batch(src, ct, dst=dst, results=res)

    def decode():
        if fse:
            hip.fse_decompress_using_dtable_batch(dst, res, dt, BLOCK, max_table_log=dec_maxlog, dst=out, results=dres)
        else:
            hip.huf_decompress4x1_using_dtable_batch(dst, res, dt, BLOCK, max_table_log=table_log, dst=out, results=dres)

    for _ in range(warmup):
        encode(); decode()
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        encode(); ev[2 * i + 1].record()
        decode(); ev[2 * i + 2].record()
    barrier()
    elapsed = time.perf_counter() - t0
    ms = (C.c_double * 16)(); launches = (C.c_uint * 16)()
    if not PLAIN:
        hip.lib.FSEHIP_probe_begin()
        for _ in range(steps):
            encode(); decode()
        barrier()
        hip.lib.FSEHIP_probe_collect(ms, launches)
    enc = sum(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(steps)) * 1e-3
    dec = sum(ev[2 * i + 1].elapsed_time(ev[2 * i + 2]) for i in range(steps)) * 1e-3
    best = min(ev[2 * i].elapsed_time(ev[2 * i + 2]) for i in range(steps)) * 1e-3
    elapsed, enc, dec, best = reduce_max([elapsed, enc, dec, best])
    assert bool((dres == BLOCK).all()) and torch.equal(out, src), "%s usingDTable(usingCTable(x)) != x" % codec_name
    parity, checked = "roundtrip-all-blocks", 0
    if rank == 0:
        lib = checker_lib()
        if lib is not None:
            checked = reference_bytes_check(lib, codec_name, table_log, src, dst, res, 0, hdr=hdr, hdr_sizes=hres)
            parity += "+%s-header-and-payload-bytes-all-%d-blocks" % (lib.kind, checked)
    total = world * nb * BLOCK * steps
    payload = float(res.sum().item()) / nb
    per = {KERNEL_NAMES[i]: (ms[i], launches[i]) for i in range(len(KERNEL_NAMES)) if launches[i]}
    rec = {"value": round(total / 2.0 ** 20 / elapsed, 1), "ms_per_step": round(elapsed / steps * 1e3, 3), "best_step_ms": round(best * 1e3, 3), "steps": steps,
           "blocks_per_gpu": nb, "encode_GBps": round(total / enc / 1e9, 2), "decode_GBps": round(total / dec / 1e9, 2),
           "payload_bytes_per_block": round(payload, 1), "header_bytes_per_block": round(float(hres.sum().item()) / nb, 1),
           "decode_maxTableLog": dec_maxlog if fse else table_log,
           "kernel_ms_per_step": {k: round(v[0] / steps, 3) for k, v in per.items()}, "parity": parity, "parity_blocks_checked": checked,
           "functions": ("FSEHIP_FSE_compress_usingCTable_batch + FSEHIP_FSE_decompress_usingDTable_batch" if fse else
                         "FSEHIP_HUF_compress4X_usingCTable_batch + FSEHIP_HUF_decompress4X1_usingDTable_batch"),
           "workload": "probagen Proba%02d, %d x 32KB blocks per GPU, tables (reference layouts, table log %d) built once by FSEHIP_%s_batch, then the "
                       "using-table calls alone" % (proba, nb, table_log, "FSE_buildCTable / FSE_buildDTable" if fse else "HUF_buildCTable / HUF_readDTableX1")}
    alg = BLOCK + payload                                          # read input once + write output once (tables excluded)
    roofs = {}
    for direction, kernels in (("encode", ("k_fse_encode_wave", "k_fse_encode") if fse else ("k_huf_encode",)), ("decode", ("k_fse_decode",) if fse else ("k_huf_decode",))):
        hot = [k for k in kernels if k in per]
        if hot:
            k = max(hot, key=lambda x: per[x][0])
            kms = per[k][0] / steps
            roofs[direction] = {"bound": "hbm", "kernel": k, "achieved": round(alg * nb / (kms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(alg * nb / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "kernel_ms_per_pass": round(kms, 3),
                                "algorithmic_bytes_per_block": round(alg, 1), "traffic": None}
            # the builder's counters of THIS call form (scripts/profile.sh runs every using-table record by itself; the caller-table FSE decoder is
            # an instantiation of its own, `k_fse_decode<true, false, true>`, and is told apart by its full name): a cross-reference like the headline's
            tp = os.path.join(ROOT, "profiles", "traffic_%s_ut_%s.json" % (k, key))
            if os.path.exists(tp):
                try:
                    trec = json.load(open(tp))
                    roofs[direction]["traffic"] = round(trec["hbm_bytes_per_block"] * nb)
                    roofs[direction]["traffic_source"] = ("profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this record alone (%s blocks per launch there), "
                                                         "HBM bytes per block scaled to this run -- recorded by the builder, not counters of this run" % (os.path.basename(tp), trec.get("blocks")))
                except Exception:
                    pass
    rec["roofline"] = roofs
    del ct, hdr, dt
    return rec


def ragged_case(hip, dev, n_blocks=20000, reps=5):
    """Per-block sizes (programs/bench.c:353-364 chunks files: the last block of every file is short; a caller's batch is ragged in
    general): n_blocks Proba14 blocks of 12,000 ... 32,768 bytes against the same number of uniform 32 KB blocks, both codecs, encode
    and decode, best of `reps`, rate per uncompressed byte.  Round trip checked on every block."""
    src = hip.probagen_batch(14, n_blocks, BLOCK, first_seed=1, device=dev)
    rng = np.random.default_rng(3)
    sizes = torch.from_numpy(rng.integers(12000, BLOCK + 1, n_blocks).astype(np.int64)).to(dev)
    total_r, total_u = float(sizes.sum().item()), float(n_blocks * BLOCK)

    def best_ms(f):
        best = None
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(); b.record(); torch.cuda.synchronize()
            t = a.elapsed_time(b)
            best = t if best is None or t < best else best
        return best

    rec = {"blocks": n_blocks, "sizes": "uniform random 12000..32768 bytes (seed 3)", "mean_bytes": round(total_r / n_blocks, 1)}
    cols = torch.arange(BLOCK, device=dev)[None, :]
    for name in ("fse", "huf"):
        comp = hip.fse_compress_batch if name == "fse" else hip.huf_compress_batch
        r = {}
        for kind, sz, tot in (("ragged", sizes, total_r), ("uniform", None, total_u)):
            cd, cr = comp(src, 11, sizes=sz)
            if name == "fse":
                out, dr = hip.fse_decompress_batch(cd, cr, BLOCK, max_log=12)
                dec = lambda: hip.fse_decompress_batch(cd, cr, BLOCK, max_log=12, dst=out, results=dr)
            else:
                dsz = sz if sz is not None else BLOCK
                out, dr = hip.huf_decompress_batch(cd, cr, dsz)
                dec = lambda: hip.huf_decompress_batch(cd, cr, dsz, dst=out, results=dr)
            torch.cuda.synchronize()
            want = sz if sz is not None else torch.full((n_blocks,), BLOCK, dtype=torch.int64, device=dev)
            coded = cr > 1
            ok = bool((dr[coded] == want[coded]).all()) and not bool((((out != src) & (cols < want[:, None])) & coded[:, None]).any())
            e_ms = best_ms(lambda: comp(src, 11, sizes=sz, dst=cd, results=cr))
            d_ms = best_ms(dec)
            r[kind] = {"encode_GBps": round(tot / e_ms / 1e6, 1), "decode_GBps": round(tot / d_ms / 1e6, 1), "encode_ms": round(e_ms, 3), "decode_ms": round(d_ms, 3),
                       "roundtrip_ok": ok, "blocks_coded": int(coded.sum().item())}
            del cd, cr, out, dr
        r["ragged_over_uniform_per_byte"] = {"encode": round(r["ragged"]["encode_GBps"] / r["uniform"]["encode_GBps"], 3),
                                             "decode": round(r["ragged"]["decode_GBps"] / r["uniform"]["decode_GBps"], 3)}
        rec[name] = r
    return rec


PLAIN = False          # --plain: see parse()


def run_case(hip, codecs, steps, warmup, barrier, rank, check=True, n_check=0):
    """time `steps` steps (each: encode + decode of every codec in `codecs`), bracketed by barrier + synchronize, with the
    kernel probe OFF; then the same steps once more with every launch of the library bracketed by HIP events on its stream
    (FSEHIP_probe_*) for the per-kernel table and the roofline.  Returns the raw timings of this rank and the per-kernel probe."""
    for _ in range(warmup):
        for cd in codecs:
            cd.encode(); cd.decode()
    barrier()
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
    # probe pass (not part of `value`)
    ms = (C.c_double * 16)(); launches = (C.c_uint * 16)()
    probe_elapsed = 0.0
    if not PLAIN:
        hip.lib.FSEHIP_probe_begin()
        t1 = time.perf_counter()
        for _ in range(steps):
            for cd in codecs:
                cd.encode(); cd.decode()
        barrier()
        probe_elapsed = time.perf_counter() - t1
        hip.lib.FSEHIP_probe_collect(ms, launches)
    enc_s = {cd.name: 0.0 for cd in codecs}; dec_s = {cd.name: 0.0 for cd in codecs}
    k = 1
    for _ in range(steps):
        for cd in codecs:
            enc_s[cd.name] += ev[k - 1].elapsed_time(ev[k]) / 1e3; k += 1
            dec_s[cd.name] += ev[k - 1].elapsed_time(ev[k]) / 1e3; k += 1
    per = {KERNEL_NAMES[i]: (ms[i], launches[i]) for i in range(len(KERNEL_NAMES)) if launches[i]}
    out = {"elapsed": elapsed, "probe_elapsed": probe_elapsed, "enc_s": enc_s, "dec_s": dec_s, "per": per, "parity": {}, "csize": {}, "parity_blocks_checked": 0,
           "step_s": [(ev[2 * len(codecs) * i].elapsed_time(ev[2 * len(codecs) * (i + 1)])) / 1e3 for i in range(steps)]}
    for cd in codecs:
        out["csize"][cd.name] = float(cd.res.sum().item()) / cd.src.shape[0]
        if check:
            out["parity"][cd.name], n = check_parity(cd, rank, n_check)
            out["parity_blocks_checked"] += n
    return out


def host_inclusive(hip, cd, chunks=8):
    """SURVEY 8(d) "report kernel-only and end-to-end (incl. H2D/D2H) separately" (protocol: programs/bench.c:349-371 times the
    call on host buffers): the headline batch with source and results in PINNED host memory -- per chunk H2D, the batched
    one-shot call, D2H of the fixed-stride result slots and sizes -- chunks pipelined over three streams (copy engines of both
    directions and the kernels overlap).  Never `value`."""
    nb = cd.src.shape[0]
    try:
        h_src = torch.empty((nb, BLOCK), dtype=torch.uint8, pin_memory=True)
        h_cmp = torch.empty((nb, cd.cap), dtype=torch.uint8, pin_memory=True)
        h_res = torch.empty(nb, dtype=torch.int64, pin_memory=True)
        h_out = torch.empty((nb, BLOCK), dtype=torch.uint8, pin_memory=True)
    except RuntimeError as e:
        return {"error": "pinned host allocation failed: %r" % (e,)}
    h_src.copy_(cd.src); torch.cuda.synchronize()
    dev = cd.src.device
    d_src, d_cmp, d_res, d_out, d_dres = torch.empty_like(cd.src), cd.dst, cd.res, cd.out, cd.dres
    s_in, s_k, s_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    edges = [nb * i // chunks for i in range(chunks + 1)]

    def pipeline(stage_in, kernels, stage_out):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(chunks):
            lo, hi = edges[i], edges[i + 1]
            with torch.cuda.stream(s_in):
                stage_in(lo, hi); e1 = torch.cuda.Event(); e1.record()
            with torch.cuda.stream(s_k):
                s_k.wait_event(e1); kernels(lo, hi); e2 = torch.cuda.Event(); e2.record()
            with torch.cuda.stream(s_out):
                s_out.wait_event(e2); stage_out(lo, hi)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def enc_in(lo, hi): d_src[lo:hi].copy_(h_src[lo:hi], non_blocking=True)
    def enc_out(lo, hi): h_cmp[lo:hi].copy_(d_cmp[lo:hi], non_blocking=True); h_res[lo:hi].copy_(d_res[lo:hi], non_blocking=True)
    def dec_in(lo, hi): d_cmp[lo:hi].copy_(h_cmp[lo:hi], non_blocking=True); d_res[lo:hi].copy_(h_res[lo:hi], non_blocking=True)
    def dec_out(lo, hi): h_out[lo:hi].copy_(d_out[lo:hi], non_blocking=True)
    if cd.name == "fse":
        def enc_k(lo, hi): hip.fse_compress_batch(d_src[lo:hi], cd.tl, dst=d_cmp[lo:hi], results=d_res[lo:hi], workspace=cd.ws_c)
        def dec_k(lo, hi): hip.fse_decompress_batch(d_cmp[lo:hi], d_res[lo:hi], BLOCK, max_log=cd.max_log, dst=d_out[lo:hi], results=d_dres[lo:hi], workspace=cd.ws_d)
    else:
        def enc_k(lo, hi): hip.huf_compress_batch(d_src[lo:hi], cd.tl, dst=d_cmp[lo:hi], results=d_res[lo:hi], workspace=cd.ws_c)
        def dec_k(lo, hi): hip.huf_decompress_batch(d_cmp[lo:hi], d_res[lo:hi], BLOCK, dst=d_out[lo:hi], results=d_dres[lo:hi], workspace=cd.ws_d)
    best_e = best_d = None
    for _ in range(3):
        te = pipeline(enc_in, enc_k, enc_out)
        td = pipeline(dec_in, dec_k, dec_out)
        best_e = te if best_e is None else min(best_e, te); best_d = td if best_d is None else min(best_d, td)
    ok = bool((h_out == h_src).all()) and bool((h_res > 1).all())
    total = nb * BLOCK
    used = float(h_res.sum().item())
    # ---- the same with PACKED results (FSEHIP_compact_batch, fsehip.h "Packed batches"): what crosses the link on the way back from the
    #      encoder / into the decoder is every block at its real size plus 8 bytes of offset per block, not fixed-stride slots.  The packed size
    #      of a chunk has to be known on the host before its copy can be issued: all chunks' uploads and kernels are queued first, then the
    #      downloads follow chunk by chunk, each behind its chunk's event.
    packed_rec = None
    try:
        flat = h_cmp.view(-1)                                                    # the pinned landing area, reused as one flat buffer
        d_packed = torch.empty(nb * BLOCK, dtype=torch.uint8, device=dev)
        d_off = [torch.empty(edges[i + 1] - edges[i] + 1, dtype=torch.int64, device=dev) for i in range(chunks)]
        h_off = [torch.empty(edges[i + 1] - edges[i] + 1, dtype=torch.int64, pin_memory=True) for i in range(chunks)]
        h_tot = torch.zeros(chunks, dtype=torch.int64, pin_memory=True)

        def enc_packed():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            evs = []
            for i in range(chunks):
                lo, hi = edges[i], edges[i + 1]
                with torch.cuda.stream(s_in):
                    enc_in(lo, hi); e1 = torch.cuda.Event(); e1.record()
                with torch.cuda.stream(s_k):
                    s_k.wait_event(e1); enc_k(lo, hi)
                    hip.compact_batch(d_cmp[lo:hi], d_res[lo:hi], d_src[lo:hi], packed=d_packed[lo * BLOCK:hi * BLOCK], offsets=d_off[i])
                    h_tot[i:i + 1].copy_(d_off[i][-1:], non_blocking=True)
                    e2 = torch.cuda.Event(); e2.record(); evs.append(e2)
            base = 0
            for i in range(chunks):
                lo, hi = edges[i], edges[i + 1]
                evs[i].synchronize()
                t = int(h_tot[i])
                with torch.cuda.stream(s_out):
                    flat[base:base + t].copy_(d_packed[lo * BLOCK:lo * BLOCK + t], non_blocking=True)
                    h_off[i].copy_(d_off[i], non_blocking=True)
                base += t
            torch.cuda.synchronize()
            return time.perf_counter() - t0, base

        def dec_packed():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            base = 0
            for i in range(chunks):
                lo, hi = edges[i], edges[i + 1]
                t = int(h_tot[i])
                with torch.cuda.stream(s_in):
                    d_packed[lo * BLOCK:lo * BLOCK + t].copy_(flat[base:base + t], non_blocking=True)
                    d_off[i].copy_(h_off[i], non_blocking=True)
                    e1 = torch.cuda.Event(); e1.record()
                with torch.cuda.stream(s_k):
                    s_k.wait_event(e1)
                    if cd.name == "fse":
                        hip.fse_decompress_packed_batch(d_packed[lo * BLOCK:hi * BLOCK], d_off[i], BLOCK, BLOCK, max_log=cd.max_log, dst=d_out[lo:hi], results=d_dres[lo:hi], workspace=cd.ws_d)
                    else:
                        hip.huf_decompress_packed_batch(d_packed[lo * BLOCK:hi * BLOCK], d_off[i], BLOCK, dst=d_out[lo:hi], results=d_dres[lo:hi], workspace=cd.ws_d)
                    e2 = torch.cuda.Event(); e2.record()
                with torch.cuda.stream(s_out):
                    s_out.wait_event(e2); dec_out(lo, hi)
                base += t
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        pe = pd = None
        pbytes = 0
        for _ in range(3):
            h_out.zero_()
            te, pbytes = enc_packed()
            td = dec_packed()
            pe = te if pe is None else min(pe, te); pd = td if pd is None else min(pd, td)
        pok = bool((h_out == h_src).all()) and pbytes == int(used)
        packed_rec = {"encode_GBps": round(total / pe / 1e9, 2), "decode_GBps": round(total / pd / 1e9, 2), "value": round(total / 2.0 ** 20 / (pe + pd), 1),
                      "encode_ms": round(pe * 1e3, 2), "decode_ms": round(pd * 1e3, 2), "roundtrip_ok": pok,
                      "pcie_bytes": {"encode": total + pbytes + 8 * (nb + chunks), "decode": pbytes + 8 * (nb + chunks) + total},
                      "what": "the same pipeline with packed results: FSEHIP_compact_batch behind the compressor, the decoder reads the packed records where they lie "
                              "(FSEHIP_*_decompress_packed_batch); the link carries every block at its real size plus 8 bytes of offset per block"}
        del d_packed, d_off
    except Exception as e:      # a second figure beside the protocol's, never a reason to lose the first
        packed_rec = {"error": repr(e)}
    return {"packed": packed_rec, "encode_GBps": round(total / best_e / 1e9, 2), "decode_GBps": round(total / best_d / 1e9, 2),
            "value": round(total / 2.0 ** 20 / (best_e + best_d), 1), "unit": "MiB/s of uncompressed data, host buffer to host buffer",
            "encode_ms": round(best_e * 1e3, 2), "decode_ms": round(best_d * 1e3, 2), "roundtrip_ok": ok,
            "pcie_bytes": {"encode": total + nb * (cd.cap + 8), "decode": nb * (cd.cap + 8) + total, "compressed_bytes_used": used},
            "what": "%d blocks in pinned host memory, %d chunks pipelined over three streams: H2D, %s_compress2 / %s_decompress batch call, D2H of "
                    "the fixed-stride result slots (%d B per block, as programs/bench.c:514-516 sizes them) and sizes; best of 3"
                    % (nb, chunks, cd.name.upper(), cd.name.upper(), cd.cap)}


def secondary_roofline(hip, cd, dev_info):
    """The bound that actually holds for k_fse_decode: blocks resident per CU (LDS capacity) x one 4-symbol iteration
    of the chain per T cycles.  T is measured by the kernel's own cycle counters (s_memtime in the decoder wave around every phase of
    16 iterations; one extra, untimed decode pass with the counters switched on), the model rate follows from it, and achieved / model
    says how much is lost outside the chain (tail of the last round of workgroups, set-up and literal tails, waits on the service waves)."""
    L = hip.lib
    if not hasattr(L, "FSEHIP_debug_decodeTiming"):
        return None
    buf = (C.c_ulonglong * 16)()
    L.FSEHIP_debug_decodeTiming(1, None)
    L.FSEHIP_probe_begin()
    cd.decode(); torch.cuda.synchronize()
    pms = (C.c_double * 16)(); pl = (C.c_uint * 16)()
    L.FSEHIP_probe_collect(pms, pl)
    L.FSEHIP_debug_decodeTiming(0, buf)
    t_run, t_wait, n_run, n_wait, n_wg, s_busy, s_idle, n_srv, clock_khz, blocks_per_wg, n_fin = [int(buf[i]) for i in range(11)]
    life_cyc, life_ticks, setup_cyc, tail_cyc, t_fin = [int(buf[i]) for i in range(11, 16)]
    wgs_per_cu, dec_waves, per_phase = max((blocks_per_wg >> 32) & 0xFF, 1), max((blocks_per_wg >> 40) & 0xFF, 1), max(blocks_per_wg >> 48, 1)
    blocks_per_wg &= 0xFFFFFFFF
    t_inner = n_srv
    if n_run == 0:
        return None
    nb = cd.src.shape[0]
    cyc_iter = t_run / (n_run * float(per_phase))
    cus = dev_info["cus"]
    resident = blocks_per_wg * wgs_per_cu
    nominal = clock_khz * 1e3
    clock = life_cyc / life_ticks * 1e8 if life_ticks else nominal          # s_memtime cycles per tick of the constant 100 MHz clock
    model_blocks_per_s = cus * resident / (BLOCK / 4.0 * cyc_iter / clock)
    ms_timed = pms[KERNEL_NAMES.index("k_fse_decode")]               # the instrumented kernel's own duration in that pass
    achieved = nb / (ms_timed * 1e-3)
    slots = cus * wgs_per_cu * dec_waves                                   # decoder waves resident on the device (each reports its lifetime)
    slot_busy = (life_ticks / 1e8) / (ms_timed * 1e-3 * slots) if life_ticks else None
    return {"bound": "chain latency x LDS-resident blocks", "kernel": "k_fse_decode",
            "resident_blocks_per_cu": resident, "cycles_per_iteration": round(cyc_iter, 1), "symbols_per_iteration": 4,
            "cycles_per_iteration_inside_the_phase": round(t_inner / (n_run * float(per_phase)), 1), "iterations_per_phase": per_phase,
            "clock_GHz": round(clock / 1e9, 3), "nominal_clock_GHz": round(nominal / 1e9, 3),
            "model_GBps": round(model_blocks_per_s * BLOCK / 1e9, 1), "achieved_GBps": round(achieved * BLOCK / 1e9, 1),
            "frac": round(achieved / model_blocks_per_s, 4),
            "where_the_rest_goes": {
                "workgroup_slots_occupied": round(slot_busy, 4) if slot_busy else None,
                "workgroup_time_in_phases": round((t_run + t_wait) / max(life_cyc, 1), 4),
                "workgroup_time_in_finishing_phases": round(t_fin / max(life_cyc, 1), 4),
                "finishing_rounds_per_workgroup": round(n_fin / max(n_wg, 1), 1), "cycles_per_finishing_round": round(t_fin / max(n_fin, 1), 1),
                "workgroup_time_setup": round(setup_cyc / max(life_cyc, 1), 4),
                "workgroup_time_literal_tail": round(tail_cyc / max(life_cyc, 1), 4),
                "decoder_wave_wait_frac": round(t_wait / max(t_run + t_wait, 1), 4),
                "service_wave_busy_frac": round(s_busy / max(s_busy + s_idle, 1), 4)},
            "note": "model = CUs x resident blocks x 4 symbols / cycles_per_iteration x clock (output bytes); cycles from s_memtime in the decoder wave around "
                    "every phase of 16 iterations, clock = those cycles per tick of the constant 100 MHz counter over the workgroups' lifetimes; one untimed "
                    "decode pass of %.2f ms with the counters on (FSEHIP_debug_decodeTiming; they cost a few per cent).  frac ~ slots occupied x time in "
                    "phases: the rest is the idle tail of the last round of workgroups, table staging / reader set-up and the literal tails" % ms_timed}



def roofline(per, codec_name, mean_csize, nb, steps, traffic_tag=None, kernels=None):
    hot = [k for k in (kernels or HOT[codec_name]) if k in per]
    if not hot:                                                   # --plain: no event probe ran
        return {"bound": "hbm", "kernel": None, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None}
    dom = max(hot, key=lambda k: per[k][0])
    dom_ms, dom_launches = per[dom]
    alg = BLOCK + mean_csize                                      # read input once + write output once
    blocks_per_launch = nb * steps / dom_launches
    achieved = alg * blocks_per_launch / (dom_ms / dom_launches * 1e-3) / 1e9
    traffic = None
    by_size = None
    traffic_source = None
    if traffic_tag is not None:
        tpath = os.path.join(ROOT, "profiles", "traffic_%s%s.json" % (dom, traffic_tag))
        if os.path.exists(tpath):
            try:   # PMC counters come from a separate rocprofv3 --pmc pass (scripts/pmc_summary.py -> profiles/); per launch like `achieved`
                rec = json.load(open(tpath))
                traffic = round(rec.get("hbm_bytes_per_block") * blocks_per_launch)
                by_size = rec.get("request_size_bytes_per_block")
                traffic_source = ("profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/profile.sh (%s blocks per launch there), HBM bytes per block "
                                  "scaled to this run's blocks per launch -- a cross-reference recorded by the builder, not counters of this run"
                                  % (os.path.basename(tpath), rec.get("blocks_per_launch", rec.get("blocks", "?"))))
            except Exception:
                traffic = None
    out = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "algorithmic_bytes_per_block": round(alg, 1),
           "blocks_per_launch": round(blocks_per_launch, 1), "avg_launch_ms": round(dom_ms / dom_launches, 4)}
    if dom.endswith("decode"):    # the decoders run one launch per class of blocks; HIP events bracket the group
        out["note"] = ("avg_launch_ms brackets the kernel's launches of one step (one per decoder class: launches over classes without blocks "
                       "return in microseconds and show up as extra calls in rocprofv3's table; profiles/*_pmc.md lists the per-pass totals)")
    if traffic_source:
        out["traffic_source"] = traffic_source
    if by_size:   # cross-check of `traffic`: the L2's memory-side requests counted by request size (exact bytes, profiles/*_pmc.md)
        out["traffic_by_request_size"] = {"read_bytes_per_block": by_size["read"], "write_bytes_per_block": by_size["write"]}
    return out


def summarize(r, codecs, nb, steps, world, reduce_max, total_blocks=None, traffic_tag=None):
    """per-configuration record from the raw timings (max over ranks)"""
    names = [cd.name for cd in codecs]
    vals = [r["elapsed"]] + [r["enc_s"][n] for n in names] + [r["dec_s"][n] for n in names]
    vals = reduce_max(vals)
    elapsed = vals[0]
    total_bytes = (total_blocks if total_blocks is not None else world * nb) * BLOCK * steps       # strong scaling: the shards add up to the corpus
    rec = {"value": round(total_bytes / 2.0 ** 20 / elapsed, 1), "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps, "blocks_per_gpu": nb}
    for i, n in enumerate(names):
        rec["%s_encode_GBps" % n] = round(total_bytes / vals[1 + i] / 1e9, 2)
        rec["%s_decode_GBps" % n] = round(total_bytes / vals[1 + len(names) + i] / 1e9, 2)
        rec["%s_compressed_bytes_per_block" % n] = round(r["csize"][n], 1)
    rec["kernel_ms_per_step"] = {k: round(v[0] / steps, 3) for k, v in r["per"].items()}
    dom_codec = max(names, key=lambda n: max(r["per"].get(k, (0, 0))[0] for k in HOT[n]))
    rec["roofline"] = roofline(r["per"], dom_codec, r["csize"][dom_codec], nb, steps, traffic_tag=traffic_tag)
    if r["parity"]:
        rec["parity"] = "; ".join("%s: %s" % (n, r["parity"][n]) for n in names)
        rec["parity_blocks_checked"] = r["parity_blocks_checked"]
    # the reference keeps the fastest of its runs (programs/bench.c:370-371); `value` is the mean over the timed steps as the harness
    # contract asks, the best step rides beside it
    best = reduce_max([min(r["step_s"])])[0]
    rec["best_step_ms"] = round(best * 1e3, 3)
    rec["best_step_value"] = round((total_blocks if total_blocks is not None else world * nb) * BLOCK / 2.0 ** 20 / best, 1)
    return rec, vals


def main():
    args = parse()
    global PLAIN
    PLAIN = args.plain
    if PLAIN:
        args.no_host_inclusive = args.no_cpu_baseline = True
    launch_ranks_if_needed(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # (FSEHIP_BENCH_BACKEND=gloo lets several ranks share one GPU for a test of the N>1 path; the driver's runs use RCCL)
    backend = os.environ.get("FSEHIP_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(minutes=8))
        else:
            dist.init_process_group(backend)
    from finitestateentropy_amd import shard
    from finitestateentropy_amd.api import FseHip, fse_compress_bound, huf_compress_bound
    hip = FseHip()
    dev = torch.device("cuda", local_rank)
    sharing = max(1, (world + max(torch.cuda.device_count(), 1) - 1) // max(torch.cuda.device_count(), 1))   # ranks per GPU (1 under the driver)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(values):
        return shard.max_over_ranks(values, dev, world)

    # N > 1: the line says which transport carried the job and how many ranks answered on it (an all-reduce of ones over the default group)
    rccl = None
    if dist is not None:
        ones = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(ones)
        rccl = {"backend": dist.get_backend(), "world": world, "ranks_seen": int(ones.item())}

    wanted = set(k for k in args.configs.split(",") if k)

    def want(key):
        return not args.no_configs and (not wanted or key in wanted)

    nb = args.blocks
    cs = args.config_steps or args.steps
    cw = 1 if args.config_steps else args.warmup
    per_block_pool = {"fse": fse_compress_bound(BLOCK) + 8 + BLOCK + 8, "huf": huf_compress_bound(BLOCK) + 8 + BLOCK + 8}
    pool_codecs = ("fse", "huf") if (args.codec == "both" or not args.no_configs) else (args.codec,)
    per_block = BLOCK + sum(per_block_pool[n] for n in pool_codecs)
    # config 5 as named: the corpus is fixed, rank r codes shard_range(total, r, world).  Everything of a rank's shard stays resident
    # (source, both codecs' slots, both outputs: 165 KB per block -> 165 GB for the whole corpus on one GPU of 288 GB); if the
    # device cannot hold that (ranks sharing a GPU in a test, a smaller part) the corpus is cut and the record says so.
    lo5, hi5 = shard.shard_range(args.cfg5_total, rank, world)
    n5s = hi5 - lo5
    corpus_note = None
    if want("cfg5_mixed_1M"):
        free_b, _ = torch.cuda.mem_get_info(dev)
        budget = int(free_b * 0.70 / sharing) - (8 << 30)
        fit = max(budget // per_block, 1024)
        fit = int(reduce_max([-float(fit)])[0] * -1)                      # the smallest budget of all ranks
        if n5s > fit:
            total_fit = fit * world
            corpus_note = "corpus cut from %d to %d blocks: %d blocks per rank is what fits the device memory free at start" % (args.cfg5_total, total_fit, fit)
            args.cfg5_total = total_fit
            lo5, hi5 = shard.shard_range(args.cfg5_total, rank, world)
            n5s = hi5 - lo5
    nmax = max([nb] + ([args.cfg5_blocks] if want("cfg5_mixed_shard") else []) + ([n5s] if want("cfg5_mixed_1M") else []))
    # device pools shared by every configuration (views are taken per case)
    pools = {}
    for n in pool_codecs:
        cap = fse_compress_bound(BLOCK) if n == "fse" else huf_compress_bound(BLOCK)
        pools["dst_" + n] = torch.empty(nmax * cap, dtype=torch.uint8, device=dev)
        pools["res_" + n] = torch.empty(nmax, dtype=torch.int64, device=dev)
        pools["out_" + n] = torch.empty(nmax * BLOCK, dtype=torch.uint8, device=dev)
        pools["dres_" + n] = torch.empty(nmax, dtype=torch.int64, device=dev)
    srcpool = torch.empty(nmax * BLOCK, dtype=torch.uint8, device=dev)

    def gen(proba, n, first_block):
        """rank-local shard of the conceptual global corpus, generated in place: global block g = first_block + row, seed g + 1"""
        view = srcpool[:n * BLOCK].view(n, BLOCK)
        if proba == "mixed":
            hip.probagen_mixed(MIX, n, BLOCK, first_block=first_block, out=view)
        else:
            hip.probagen_batch(proba, n, BLOCK, first_seed=1 + first_block, out=view)
        return view

    # ---------------- headline: configs[1] per rank
    head_proba = "mixed" if args.workload == "mixed" else args.proba
    src = gen(head_proba, nb, rank * nb)
    head_names = ("fse", "huf") if args.codec == "both" else (args.codec,)
    codecs = [Codec(hip, n, src, pools, args.table_log, args.max_log) for n in head_names]
    r = run_case(hip, codecs, args.steps, args.warmup, barrier, rank, n_check=args.parity_blocks)
    head, vals = summarize(r, codecs, nb, args.steps, world, reduce_max)
    elapsed = vals[0]
    dom_codec = max(head_names, key=lambda n: max(r["per"].get(k, (0, 0))[0] for k in HOT[n]))
    head_roof = roofline(r["per"], dom_codec, r["csize"][dom_codec], nb, args.steps, traffic_tag="")
    enc_roof = roofline(r["per"], dom_codec, r["csize"][dom_codec], nb, args.steps, traffic_tag="",
                        kernels=[k for k in HOT[dom_codec] if "encode" in k])
    head_roof["measured_over"] = ("%d steps repeated with the event probe on, directly after the timed region (%.3f ms per step with the probe, %.3f without)"
                                  % (args.steps, r["probe_elapsed"] / args.steps * 1e3, r["elapsed"] / args.steps * 1e3))
    if rank == 0 and dom_codec == "fse" and not PLAIN:
        try:
            sec = secondary_roofline(hip, codecs[head_names.index("fse")], hip.device_info())
            if sec:
                head_roof["secondary"] = sec
        except Exception as e:
            head_roof["secondary"] = {"error": repr(e)}
    host_inc = None
    if world == 1 and not args.no_host_inclusive and head_proba != "mixed":
        try:
            host_inc = host_inclusive(hip, codecs[0])
        except Exception as e:
            host_inc = {"error": repr(e)}
    cpu_sample = None
    if world == 1 and not args.no_cpu_baseline and head_proba != "mixed":
        cpu_sample = src[:min(args.cpu_sample_blocks, nb)].cpu().numpy()
    del codecs

    configs = {}
    comm_leg = None
    if not args.no_configs:

        def case(key, proba, names, n, first_block, table_log=args.table_log, max_log=args.max_log, desc="", total=None, n_check=None, traffic_tag=None):
            s = gen(proba, n, first_block)
            cds = [Codec(hip, nm, s, pools, table_log, max_log) for nm in names]
            rr = run_case(hip, cds, cs, cw, barrier, rank, n_check=args.parity_blocks if n_check is None else n_check)
            rec, _ = summarize(rr, cds, n, cs, world, reduce_max, total_blocks=total, traffic_tag=traffic_tag)
            rec["workload"] = desc
            configs[key] = rec
            return s, cds

        if want("fse_maxlog11"):
            case("fse_maxlog11", args.proba, ("fse",), nb, rank * nb, max_log=max(args.table_log, 9),
                 desc="headline workload decoded with FSE_decompress_wksp(maxLog = 11) (caller promises tableLog <= 11)")
        if want("cfg3_p80_fse"):
            case("cfg3_p80_fse", 80, ("fse",), nb, rank * nb, traffic_tag="_p80", desc="BASELINE configs[2]: probagen Proba80, %d x 32KB blocks per GPU, FSE encode+decode" % nb)
        if want("cfg4_p14_huf"):
            case("cfg4_p14_huf", 14, ("huf",), nb, rank * nb, traffic_tag="", desc="BASELINE configs[3]: probagen Proba14, %d x 32KB blocks per GPU, Huff0 4-stream encode + HUF_decompress" % nb)
        if want("fse_tl12"):
            case("fse_tl12", 14, ("fse",), nb, rank * nb, table_log=12, desc="Proba14, FSE with tableLog 12 (what `fse -b` requests, programs/bench.c:113)")
        if want("huf_tl12"):
            case("huf_tl12", 2, ("huf",), nb, rank * nb, table_log=12, desc="Proba02 (256 symbols), Huff0 with tableLog 12 (HUF_TABLELOG_MAX)")
        if want("using_tables"):
            ut = {}
            ut_keys = set(k for k in args.ut_keys.split(",") if k)
            for key, codec_name, proba in (("fse_p14", "fse", 14), ("fse_p80", "fse", 80), ("huf_p14", "huf", 14)):
                if ut_keys and key not in ut_keys:
                    continue
                s_ut = gen(proba, nb, rank * nb)
                ut[key] = using_tables_case(hip, codec_name, proba, s_ut, pools, args.table_log, cs, cw, barrier, reduce_max, world, rank, key)
            ut["note"] = ("the north-star-named calls on caller-built tables in the reference's layouts; FSE decode converts the tables to its bit-reversed "
                          "cells while staging them and runs the same fast loop as the one-shot path (compare kernel_ms_per_step.k_fse_decode with the "
                          "headline's and cfg3's)")
            configs["using_tables"] = ut
        if want("cfg5_mixed_shard"):
            n5 = args.cfg5_blocks
            s5, cds5 = case("cfg5_mixed_shard", "mixed", ("fse", "huf"), n5, rank * n5, n_check=args.cfg5_parity_blocks, traffic_tag="_mixed",
                            desc="BASELINE configs[4]'s mix at a fixed %d x 32KB blocks per GPU (%d in all): probagen mixed {P02,P14,P80} (block g: P[g mod 3], "
                                 "seed g+1), FSE and Huff0 round trip of every block, contiguous block ranges, no collective" % (n5, n5 * world))
            configs["cfg5_mixed_shard"]["scaling"] = "weak"
            configs["cfg5_mixed_shard"]["value_note"] = "value counts each block once per step although it goes through both codecs"
            del s5, cds5
        if want("cfg5_mixed_1M"):
            total = args.cfg5_total
            s5, cds5 = case("cfg5_mixed_1M", "mixed", ("fse", "huf"), n5s, lo5, total=total, n_check=args.cfg5_parity_blocks, traffic_tag="_mixed",
                            desc="BASELINE configs[4] as named: probagen mixed {P02,P14,P80} (block g: P[g mod 3], seed g+1), %d x 32KB blocks in all, "
                                 "FSE and Huff0 round trip of every block, rank r codes shard_range(%d, r, %d) (this run: %d blocks per GPU); compute-only "
                                 "(each rank generates its shard, no collective)" % (total, total, world, n5s))
            rec5 = configs["cfg5_mixed_1M"]
            rec5["scaling"] = "strong"; rec5["corpus_blocks"] = total
            rec5["value_note"] = "value counts each block once per step although it goes through both codecs"
            if corpus_note:
                rec5["corpus_note"] = corpus_note
            if world > 1:
                comm_leg = (total, cds5)              # runs LAST, under a watchdog (below): a transfer that never completes must not cost the line
            del s5, cds5
        if want("ragged") and world == 1 and not PLAIN:
            try:
                configs["ragged"] = ragged_case(hip, dev)
            except Exception as e:          # a report beside the BASELINE configurations, never a reason to lose the line
                configs["ragged"] = {"error": repr(e)}
        if want("fse_u16") and hasattr(hip, "fse_compress_u16_batch"):
            configs["fse_u16"] = u16_case(hip, dev, args.u16_blocks, cs, barrier, reduce_max, world, rank)

    # ---- the communication leg of config 5 (N > 1), after everything else has been measured and the line is ready: the scatter / gather over
    #      RCCL is the one part of this program no box of ours has ever run across GPUs.  A watchdog on every rank bounds it: if the leg has
    #      not finished after --comm-timeout seconds, rank 0 prints the line with an error record in its place and every rank leaves through
    #      os._exit (a hung collective cannot be cancelled from Python); the NCCL watchdog of the process group (8 minutes) comes later.
    def comm_leg_guarded(emit_without):
        import threading
        done = threading.Event()

        def give_up():
            if done.is_set():
                return
            sys.stderr.write("bench.py rank %d: the communication leg did not finish within %d s -- leaving\n" % (rank, args.comm_timeout)); sys.stderr.flush()
            if rank == 0:
                emit_without("the communication leg did not finish within %d s (a transfer or a collective never completed); every other figure of this line was measured before it" % args.comm_timeout)
            os._exit(0)
        timer = threading.Timer(float(args.comm_timeout), give_up)
        timer.daemon = True
        # every rank arms its watchdog at the same moment: rank 0 gets here after its CPU baseline (tens of seconds), the others at once -- they
        # wait for it in a collective first, so that their timers do not count the root's host work as communication time (rank 0 arms its
        # own timer in front of that collective: a transport that cannot even do a barrier must not cost the line either)
        if rank == 0:
            timer.start()
        if dist is not None:
            dist.barrier()
        if rank != 0:
            timer.start()
        total5, cds = comm_leg
        try:
            rec = with_comm_case(args, hip, shard, dev, rank, world, total5, cds, srcpool, barrier, reduce_max, sharing)
        except Exception as e:       # the communication leg is a report beside the compute-only record, never a reason to lose the line
            rec = {"value": None, "error": repr(e)}
        done.set(); timer.cancel()
        return rec

    if rank != 0:
        if comm_leg is not None:
            comm_leg_guarded(None)
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
        "roofline": head_roof, "roofline_encode": enc_roof,
        "kernel_ms_per_step": head["kernel_ms_per_step"],
    }
    if rccl is not None:
        line["rccl"] = rccl
    if len(head_names) > 1:
        for n in head_names[1:]:
            line["%s_encode_GBps" % n] = head["%s_encode_GBps" % n]; line["%s_decode_GBps" % n] = head["%s_decode_GBps" % n]
    if host_inc is not None:
        line["host_inclusive"] = host_inc
    if configs:
        line["configs"] = configs
    if cpu_sample is not None:
        try:
            line["cpu_baseline"] = cpu_baseline(args, cpu_sample, head_names[0])
        except Exception as e:  # the baseline is a report, never a reason to lose the GPU line
            line["cpu_baseline"] = {"value": None, "unit": "MiB/s", "cores": host_threads(), "kind": "unavailable", "sample": repr(e)}
    def emit():
        """detail file first, then the one short line (the last thing on stdout)"""
        extra = {}
        if rccl is not None:
            extra["rccl"] = rccl
            rec5 = configs.get("cfg5_mixed_1M")
            if rec5:
                extra["compute_only"] = {"value": rec5["value"], "ms_per_step": rec5["ms_per_step"]}
                wc = rec5.get("with_comm")
                if wc:
                    pp = wc.get("pipelined_packed") or {}
                    links = wc.get("root_link_GBps") or {}
                    extra["with_comm"] = {"serial_ms": wc.get("ms"), "pipelined_ms": pp.get("ms"), "scatter_GBps": links.get("scatter"),
                                          "gather_GBps": links.get("gather"), "roundtrip_ok": wc.get("roundtrip_ok")}
                    if wc.get("error"):
                        extra["with_comm"]["error"] = wc["error"][:160]
        print(compact_line(line, extra, write_detail(line)))
        sys.stdout.flush()

    if comm_leg is not None:
        def emit_without(why):
            configs["cfg5_mixed_1M"]["with_comm"] = {"value": None, "error": why}
            emit()
        configs["cfg5_mixed_1M"]["with_comm"] = comm_leg_guarded(emit_without)
    emit()
    if dist is not None:
        dist.destroy_process_group()


def with_comm_case(args, hip, shard, dev, rank, world, total, cds5, srcpool, barrier, reduce_max, sharing):
    """config 5 with the corpus on ONE rank (north_star: "RCCL broadcast/gather over xGMI only for the block scatter/gather"): rank 0
    holds the `total` blocks; grouped scatter of the raw blocks, FSE + Huff0 round trip of every rank's shard, grouped gather of
    both codecs' fixed-stride compressed slots and sizes (finitestateentropy_amd/shard.py: one ncclGroup per direction, a star
    over the root's xGMI links).  One untimed pass first (RCCL creates its communicators and channels on first use), then
    `--comm-passes` timed passes, each bracketed by barriers, max over ranks."""
    lo, hi = shard.shard_range(total, rank, world)
    n = hi - lo
    corpus, gather_out = None, None
    need = total * (BLOCK + sum(cd.cap + 8 for cd in cds5))
    ok_mem = True
    if rank == 0:
        free_b, _ = torch.cuda.mem_get_info(dev)
        ok_mem = need < free_b * 0.85 / sharing
    ok_mem = reduce_max([0.0 if ok_mem else 1.0])[0] == 0.0
    if not ok_mem:
        return {"value": None, "error": "the root cannot hold the corpus and the gathered slots (%d bytes) beside its shard" % need}
    if rank == 0:
        corpus = torch.empty((total, BLOCK), dtype=torch.uint8, device=dev)
        hip.probagen_mixed(MIX, total, BLOCK, first_block=0, out=corpus)
        gather_out = [(torch.empty((total, cd.cap), dtype=torch.uint8, device=dev), torch.empty(total, dtype=torch.int64, device=dev)) for cd in cds5]
    shard_out = srcpool[:n * BLOCK].view(n, BLOCK)
    times, phases = [], []
    ok = True
    for p in range(args.comm_passes + 1):
        marks = []

        def mark(name):
            torch.cuda.synchronize(); marks.append((name, time.perf_counter()))
        barrier()
        t0 = time.perf_counter()
        mine, gathered = shard.sharded_codec_job(corpus, total, BLOCK, rank, world, dev, cds5, shard_out=shard_out, gather_out=gather_out, mark=mark)
        barrier()
        t = reduce_max([time.perf_counter() - t0])[0]
        if p == 0:
            ok = shard.sharded_job_ok(mine, gathered, cds5, total, BLOCK, rank, world)
            continue
        times.append(t)
        prev = t0; ph = {}
        for name, tm in marks:
            ph[name] = tm - prev; prev = tm
        keys = sorted(ph)
        mx = reduce_max([ph[k] for k in keys])
        phases.append(dict(zip(keys, mx)))
    okv = reduce_max([0.0 if ok else 1.0])[0] == 0.0
    # ---- the same job pipelined and with variable-length results (shard.sharded_codec_job_pipelined): every shard in `--comm-pieces`
    #      pieces -- scatter of piece k + 1, codecs of piece k and gather of piece k - 1 overlap -- and what travels back is the packed
    #      records of FSEHIP_compact_batch (every block at its real size) plus their offsets instead of fixed-stride slots
    pipe = None
    try:
        def compact_fn(pc, piece_src):
            return hip.compact_batch(pc.dst, pc.res, piece_src)
        packed_out = offsets_out = None
        if rank == 0:
            packed_out = [torch.empty(total * BLOCK, dtype=torch.uint8, device=dev) for _ in cds5]
            offsets_out = [torch.empty(total + 1, dtype=torch.int64, device=dev) for _ in cds5]
        ptimes, stats, pok = [], None, True
        # a communicator of its own for the way back: RCCL runs the operations of one communicator on one stream, in order -- with one per
        # direction the scatter of piece k + 1 and the gather of piece k - 1 are in flight together (created once, collectively)
        import torch.distributed as dist
        back_group = dist.new_group() if world > 1 and dist.get_backend() == "nccl" else None
        for p in range(args.comm_passes + 1):
            barrier()
            t0 = time.perf_counter()
            mine, packed_g, stats = shard.sharded_codec_job_pipelined(corpus, total, BLOCK, rank, world, dev, cds5, compact_fn, pieces=args.comm_pieces,
                                                                      shard_out=shard_out, packed_out=packed_out, offsets_out=offsets_out, gather_group=back_group)
            barrier()
            t = reduce_max([time.perf_counter() - t0])[0]
            if p == 0:                                            # untimed pass: every rank's round trip, and the root decodes what it gathered
                pok = all(bool((cd.dres == BLOCK).all()) and torch.equal(cd.out, mine) for cd in cds5)
                if rank == 0:
                    order = torch.tensor(stats["order"], device=dev)
                    for cd, (pk, of) in zip(cds5, packed_g):
                        for a0 in range(0, total, 65536):         # the packed stream decoded where it lies, 64k records at a time
                            a1 = min(a0 + 65536, total)
                            sub = of[a0:a1 + 1]
                            if cd.name == "fse":
                                o, r = hip.fse_decompress_packed_batch(pk, sub, BLOCK, BLOCK, max_log=cd.max_log)
                            else:
                                o, r = hip.huf_decompress_packed_batch(pk, sub, BLOCK)
                            pok = pok and bool((r == BLOCK).all()) and torch.equal(o, corpus[order[a0:a1]])
                            del o, r
                continue
            ptimes.append(t)
        pokv = reduce_max([0.0 if pok else 1.0])[0] == 0.0
        pmean = sum(ptimes) / len(ptimes)
        pipe = {"value": round(total * BLOCK / 2.0 ** 20 / pmean, 1), "ms": round(pmean * 1e3, 2), "best_ms": round(min(ptimes) * 1e3, 2), "passes": len(ptimes),
                "pieces_per_shard": args.comm_pieces, "roundtrip_ok": bool(pokv),
                "scatter_bytes": stats["scatter_bytes"], "gather_bytes": stats["gather_bytes"], "payload_bytes": stats["payload_bytes"],
                "fixed_stride_gather_bytes": (total - n) * sum(cd.cap + 8 for cd in cds5) if rank == 0 else 0,
                "what": "the same job in %d pieces per shard: scatter of piece k + 1, codecs of piece k and gather of piece k - 1 overlap -- transfers are posted "
                        "and waited for on two side streams (one per direction, each direction a communicator of its own), the compute stream waits only for the "
                        "arrival of its next piece and joins the side streams once at the end, the packed sizes travel as host integers over a gloo side "
                        "group; the gather ships the packed records of FSEHIP_compact_batch -- every block at its real size, sizes exchanged first -- "
                        "plus 8 bytes per record offset; on the untimed pass the root decodes the gathered packed streams where they lie and compares them "
                        "with the corpus" % args.comm_pieces}
        del packed_out, offsets_out
    except Exception as e:
        pipe = {"value": None, "error": repr(e)}
    mean = sum(times) / len(times)
    avg_ph = {k: round(sum(p_[k] for p_ in phases) / len(phases) * 1e3, 2) for k in phases[0]}
    scatter_bytes = (total - n) * BLOCK if rank == 0 else 0
    gather_bytes = (total - n) * sum(cd.cap + 8 for cd in cds5) if rank == 0 else 0
    rec = {"value": round(total * BLOCK / 2.0 ** 20 / mean, 1), "unit": "MiB/s of uncompressed data (each block counted once; it goes through both codecs)",
           "ms": round(mean * 1e3, 2), "best_ms": round(min(times) * 1e3, 2), "passes": len(times), "roundtrip_ok": bool(okv),
           "phase_ms": avg_ph,
           "root_link_GBps": {"scatter": round(scatter_bytes / max(avg_ph.get("1_scatter", 0) * 1e-3, 1e-9) / 1e9, 1) if scatter_bytes else None,
                              "gather": round(gather_bytes / max(avg_ph.get("3_gather", 0) * 1e-3, 1e-9) / 1e9, 1) if gather_bytes else None},
           "what": "rank 0 holds %d blocks: grouped scatter (one transfer per peer) + FSE and Huff0 encode+decode of every shard + grouped gather of both "
                   "codecs' fixed-stride compressed slots and sizes; 1 untimed pass (communicator set-up), then %d timed passes; value from their mean; "
                   "phase_ms = max over ranks of each phase, measured with a device synchronize between phases" % (total, len(times))}
    rec["pipelined_packed"] = pipe
    del corpus, gather_out
    return rec

ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_block", "avg_launch_ms")
LINE_LIMIT = 4096          # the driver reads the LAST stdout line out of an 8 KB tail; round 5's 21 KB line was lost that way


def compact_line(full, extra=None, detail_path=None):
    """The ONE line rank 0 prints: the contract fields, one roofline per direction, the CPU baseline and a {value, ms_per_step, frac}
    triple per BASELINE configuration (protocol: one short result line per run, programs/bench.c:458-468).  Everything else -- the
    secondary roofline, host-inclusive figures, using-table calls, tableLog-12 and 16-bit records, notes -- is in the detail file."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: full[k] for k in keep}
    cfg = full["config"]
    line["config"] = {k: cfg[k] for k in ("workload", "blocks_per_gpu", "block_bytes", "codec", "parity") if k in cfg}
    line["encode_GBps"], line["decode_GBps"] = full["encode_GBps"], full["decode_GBps"]
    line["roofline"] = {k: full["roofline"].get(k) for k in ROOF_KEYS}
    if full.get("roofline_encode"):
        line["roofline_encode"] = {k: full["roofline_encode"].get(k) for k in ROOF_KEYS}
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "cpu_model", "encode_MiBps", "decode_MiBps", "single_thread_value") if k in cb}
    cfgs = {}
    for key, rec in (full.get("configs") or {}).items():
        if not key.startswith("cfg") or not isinstance(rec, dict) or "value" not in rec:
            continue
        roof = rec.get("roofline") or {}
        cfgs[key] = {"value": rec["value"], "ms_per_step": rec["ms_per_step"], "frac": roof.get("frac"), "kernel": roof.get("kernel"),
                     "parity_blocks_checked": rec.get("parity_blocks_checked")}
    if cfgs:
        line["configs"] = cfgs
    for k, v in (extra or {}).items():
        line[k] = v
    line["detail"] = detail_path
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:        # never lose the line to its own size: drop the optional parts, longest first
        for k in ("configs", "roofline_encode", "with_comm", "compute_only"):
            line.pop(k, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) < LINE_LIMIT:
                break
    return text


def write_detail(full):
    """bench_detail.json beside bench.py (and under gpurun_out/ when that directory exists, so that a gpurun call brings it back)"""
    path = None
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if d != ROOT and not os.path.isdir(d):
            continue
        try:
            with open(os.path.join(d, "bench_detail.json"), "w") as f:
                json.dump(full, f, indent=1)
            path = path or os.path.relpath(os.path.join(d, "bench_detail.json"), ROOT)
        except OSError:
            pass
    return path


if __name__ == "__main__":
    main()
