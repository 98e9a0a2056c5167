#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X block entropy codec (contract: see task statement).

A "step" = one pass of the hot path over one batch of synthetic probagen blocks resident in HBM:
FSE encode (FSE_compress2 semantics: histogram, normalisation, NCount header, CTable, payload) followed
by FSE decode (FSE_decompress) of every block.  N=1 workload = BASELINE.json configs[1]:
"probagen Proba14, 100k x 32KB blocks, FSE encode+decode on 1xMI355X, bit-exact check".

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `value` = uncompressed MiB that went through encode AND decode per second
(whole job, all ranks), inputs resident in HBM.  Blocks shard across ranks with no data-path collective
(every block is independent: programs/bench.c:353-364) -> "scaling": "weak" (100k blocks per rank).
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
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=100000, help="32 KB blocks per GPU")
    ap.add_argument("--proba", type=int, default=14)
    ap.add_argument("--codec", choices=["fse", "huf"], default="fse")
    ap.add_argument("--table-log", type=int, default=11)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-blocks", type=int, default=20000)
    return ap.parse_args()


def cpu_baseline(args):
    """The reference (oracle/_ref, kind 'reference') or our port (oracle/liboracle.so, kind 'port') on this
    box's host cores, same workload shape, bounded sample; encode+decode round trip, all cores (OpenMP)."""
    from oracle.oracle import Oracle, Ref
    orc = Oracle()
    lib = Ref() if Ref.available() else orc
    n = args.cpu_sample_blocks
    src = orc.probagen_batch(args.proba, n, BLOCK, 1)
    codec = 0 if args.codec == "fse" else 1
    cores = os.cpu_count() or 1
    lib.compress_batch(codec, src[:64], table_log=args.table_log, nthreads=cores)      # warm
    t_enc, res, comp = lib.compress_batch(codec, src, table_log=args.table_log, nthreads=cores)
    t_dec, dres, out = lib.decompress_batch(codec, comp, res, BLOCK, nthreads=cores)
    assert (dres == BLOCK).all() and (out == src).all()
    t1e, _, _ = lib.compress_batch(codec, src[:n // 8], table_log=args.table_log, nthreads=1)
    t1d, _, _ = lib.decompress_batch(codec, comp[:n // 8], res[:n // 8], BLOCK, nthreads=1)
    mib = n * BLOCK / 2.0 ** 20
    return {
        "value": round(mib / (t_enc + t_dec), 1), "unit": "MiB/s (encode+decode round trip, uncompressed bytes)",
        "cores": cores, "kind": lib.kind,
        "sample": "%d probagen P%02d blocks of 32 KB, %s_compress2 + %s_decompress, OpenMP over blocks" % (n, args.proba, args.codec.upper(), args.codec.upper()),
        "encode_MiBps": round(mib / t_enc, 1), "decode_MiBps": round(mib / t_dec, 1),
        "single_thread_encode_MiBps": round(mib / 8 / t1e, 1), "single_thread_decode_MiBps": round(mib / 8 / t1d, 1),
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from finitestateentropy_amd.api import FseHip, fse_compress_bound, huf_compress_bound
    hip = FseHip()
    dev = torch.device("cuda", local_rank)

    nb = args.blocks
    # shard: rank r owns blocks [r*nb, (r+1)*nb) of the (conceptual) global batch; block b uses seed b+1
    src = hip.probagen_batch(args.proba, nb, BLOCK, first_seed=1 + rank * nb, device=dev)
    cap = fse_compress_bound(BLOCK) if args.codec == "fse" else huf_compress_bound(BLOCK)
    dst = torch.empty((nb, cap), dtype=torch.uint8, device=dev)
    res = torch.empty(nb, dtype=torch.int64, device=dev)
    out = torch.empty((nb, BLOCK), dtype=torch.uint8, device=dev)
    dres = torch.empty(nb, dtype=torch.int64, device=dev)
    if args.codec == "fse":
        ws_c = hip.fse_workspace(nb, args.table_log, False, dev)
        ws_d = hip.fse_workspace(nb, args.table_log, True, dev)

        def encode():
            hip.fse_compress_batch(src, args.table_log, dst=dst, results=res, workspace=ws_c)

        def decode():
            hip.fse_decompress_batch(dst, res, BLOCK, max_log=max(args.table_log, 9), dst=out, results=dres, workspace=ws_d)
        hot = ("k_fse_encode", "k_fse_encode_wave", "k_fse_decode")
    else:
        ws_c = hip.huf_workspace(nb, False, dev)
        ws_d = hip.huf_workspace(nb, True, dev)

        def encode():
            hip.huf_compress_batch(src, args.table_log, dst=dst, results=res, workspace=ws_c)

        def decode():
            hip.huf_decompress_batch(dst, res, BLOCK, dst=out, results=dres, workspace=ws_d)
        hot = ("k_huf_encode", "k_huf_decode")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        encode(); decode()
    barrier()
    hip.lib.FSEHIP_probe_begin()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for k in range(args.steps):
        encode(); ev[2 * k + 1].record()
        decode(); ev[2 * k + 2].record()
    barrier()
    elapsed = time.perf_counter() - t0
    ms = (C.c_double * 16)(); launches = (C.c_uint * 16)()
    hip.lib.FSEHIP_probe_collect(ms, launches)
    enc_s = sum(ev[2 * k].elapsed_time(ev[2 * k + 1]) for k in range(args.steps)) / 1e3
    dec_s = sum(ev[2 * k + 1].elapsed_time(ev[2 * k + 2]) for k in range(args.steps)) / 1e3

    # ---- bit-exact gates (untimed): round trip on every block; encoder bytes vs the CPU oracle on a sample
    assert bool((dres == BLOCK).all()), "decode return values wrong"
    assert torch.equal(out, src), "decode(encode(x)) != x"
    csum = int(res.sum().item())
    parity = "roundtrip-all-blocks"
    if rank == 0:
        try:
            from oracle.oracle import Oracle
            orc = Oracle()
            m = min(256, nb)
            host = src[:m].cpu().numpy()
            _, ores, odst = orc.compress_batch(0 if args.codec == "fse" else 1, host, table_log=args.table_log)
            rh, dh = res[:m].cpu().numpy(), dst[:m].cpu().numpy()
            assert (rh == ores.astype(np.int64)).all(), "encode sizes differ from the CPU oracle"
            for b in range(m):
                assert (dh[b][:rh[b]] == odst[b][:rh[b]]).all(), "encode bytes differ from the CPU oracle (block %d)" % b
            parity += "+oracle-bytes-%d-blocks" % m
        except OSError:
            parity += "(oracle unavailable)"

    # max over ranks
    t = torch.tensor([elapsed, enc_s, dec_s], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, enc_s, dec_s = [float(x) for x in t.tolist()]
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_mib = world * nb * BLOCK * args.steps / 2.0 ** 20
    mean_csize = csum / nb
    # dominant kernel of the step, from the live HIP-event probe
    per = {KERNEL_NAMES[i]: (ms[i], launches[i]) for i in range(len(KERNEL_NAMES)) if launches[i]}
    dom = max(hot, key=lambda k: per.get(k, (0, 0))[0])
    dom_ms, dom_launches = per[dom]
    alg_bytes_per_block = BLOCK + mean_csize                      # read input once + write output once
    blocks_per_launch = nb * args.steps / dom_launches
    achieved = alg_bytes_per_block * blocks_per_launch / (dom_ms / dom_launches * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % dom)
    if os.path.exists(tpath):
        try:
            # PMC counters come from a separate rocprofv3 --pmc pass (scripts/pmc_summary.py -> profiles/); per launch like `achieved`
            traffic = round(json.load(open(tpath)).get("hbm_bytes_per_block") * blocks_per_launch)
        except Exception:
            traffic = None
    line = {
        "metric": "MB/s encode+decode per GPU on 32KB probagen blocks; bit-exact vs CPU ref",
        "value": round(total_mib / elapsed, 1), "unit": "MiB/s of uncompressed data through encode+decode (whole job)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "probagen Proba%02d, %d x 32KB blocks per GPU, %s encode+decode, tableLog %d, bit-exact check"
                   % (args.proba, nb, "FSE" if args.codec == "fse" else "Huff0 4-stream", args.table_log),
                   "blocks_per_gpu": nb, "block_bytes": BLOCK, "codec": args.codec, "parity": parity,
                   "compressed_bytes_per_block": round(mean_csize, 1)},
        "encode_GBps": round(world * nb * BLOCK * args.steps / enc_s / 1e9, 2),
        "decode_GBps": round(world * nb * BLOCK * args.steps / dec_s / 1e9, 2),
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "algorithmic_bytes_per_block": round(alg_bytes_per_block, 1),
                     "blocks_per_launch": round(blocks_per_launch, 1), "avg_launch_ms": round(dom_ms / dom_launches, 4)},
        "kernel_ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in per.items()},
    }
    if world == 1 and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline(args)
        except Exception as e:  # the baseline is a report, never a reason to lose the GPU line
            line["cpu_baseline"] = {"value": None, "unit": "MiB/s", "cores": os.cpu_count(), "kind": "unavailable", "sample": repr(e)}
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
