"""Turn rocprofv3 outputs (gpurun_out/<run>/{trace,pmc_fetch,pmc_write}) into the committed summaries under profiles/.

usage: python scripts/pmc_summary.py gpurun_out/r1 r01 <blocks_in_pmc_run> [traffic-file suffix, e.g. _p80]
Writes profiles/<tag>_kernel_stats.csv (copy of rocprofv3 --kernel-trace --stats), profiles/<tag>_pmc.md and
profiles/traffic_<kernel>.json (HBM bytes per block, read by bench.py for roofline.traffic).

HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB, collected in separate --pmc passes;
on gfx950 FETCH_SIZE reports half of the bytes of a coalesced streaming read, so it is doubled.  The correction is
calibrated in the same run on k_hist, whose read volume is known exactly (every source byte once).
"""
import collections
import csv
import json
import os
import shutil
import sys


def agg(path, counter=None):
    out = collections.defaultdict(lambda: [0, 0.0, 0])
    if not os.path.exists(path):
        return out
    for r in csv.DictReader(open(path)):
        if counter is not None and r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("void "):
            k = k[5:]
        k = k.split("<")[0]                       # template instances (k_fse_decode<true>) report under the kernel name
        a = out[k]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += int(r["Grid_Size"])
    return out


def ut_summary(run_prefix, tag, nblocks):
    """python scripts/pmc_summary.py --ut gpurun_out/<run> <tag> <blocks>: the using-table records (scripts/profile.sh: one record per run, in
    <run>_ut_<key>/pmc_fetch|pmc_write).  Kernels are told apart by their FULL names -- the caller-table FSE decoder is `k_fse_decode<true, false,
    true>`, the one-shot decoder of the same run's headline `<true, false, false>` -- and a kernel's bytes are divided by its launches that carried
    blocks (3 per record: warm-up + 2 steps; the encoder is launched by the headline of the run as well: 6)."""
    lines = ["# %s: HBM traffic of the using-table calls (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one record per run)" % tag, "",
             "| record | kernel | launches counted | fetch KiB (raw) | write KiB | HBM bytes / block (FETCH x2) |", "|---|---|---|---|---|---|"]
    want = {"fse_p14": (("k_fse_decode", "<true, false, true>", 3), ("k_fse_encode_wave", "", 6)),
            "fse_p80": (("k_fse_decode", "<true, false, true>", 3), ("k_fse_encode_wave", "", 6)),
            "huf_p14": (("k_huf_decode", "k_huf_decode", 3), ("k_huf_encode", "", 3))}
    for key, kernels in want.items():
        run = "%s_ut_%s" % (run_prefix, key)
        sums = {}
        for sub in ("pmc_fetch", "pmc_write"):
            path = os.path.join(run, sub, "bench_counter_collection.csv")
            if not os.path.exists(path):
                continue
            for r in csv.DictReader(open(path)):
                name = r["Kernel_Name"]
                for k, must, passes in kernels:
                    base = name.split("(")[0].replace("void ", "")
                    if k == "k_huf_decode":
                        ok = base.startswith("k_huf_decode")            # the stream-parallel launches and the serial one: the whole a5 step
                    else:
                        ok = base.split("<")[0] == k and must in name
                    if ok:
                        sums.setdefault(k, {"pmc_fetch": 0.0, "pmc_write": 0.0})[sub] += float(r["Counter_Value"])
        for k, must, passes in kernels:
            if k not in sums:
                continue
            f, w = sums[k]["pmc_fetch"], sums[k]["pmc_write"]
            per_block = (f * 1024 * 2.0 + w * 1024) / (nblocks * passes)
            lines.append("| %s | %s%s | %d | %.0f | %.0f | %.0f |" % (key, k, must if must.startswith("<") else "", passes, f, w, per_block))
            json.dump({"kernel": k, "record": key, "hbm_bytes_per_block": round(per_block, 1), "fetch_correction": 2.0, "fetch_KiB_raw_total": f, "write_KiB_total": w,
                       "blocks": nblocks, "passes": passes, "source": tag}, open("profiles/traffic_%s_ut_%s.json" % (k, key), "w"))
    open("profiles/%s_pmc.md" % tag, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


def main():
    if sys.argv[1] == "--ut":
        return ut_summary(sys.argv[2], sys.argv[3], int(sys.argv[4]))
    run, tag, nblocks = sys.argv[1], sys.argv[2], int(sys.argv[3])
    suffix = sys.argv[4] if len(sys.argv) > 4 else ""             # traffic_<kernel><suffix>.json: records of a configuration other than the headline's
    os.makedirs("profiles", exist_ok=True)
    shutil.copy(os.path.join(run, "trace", "bench_kernel_stats.csv"), "profiles/%s_kernel_stats.csv" % tag)
    fetch = agg(os.path.join(run, "pmc_fetch", "bench_counter_collection.csv"))
    write = agg(os.path.join(run, "pmc_write", "bench_counter_collection.csv"))
    # exact byte counts from the L2's memory-side request counters by request size (cross-check, no calibration needed)
    rdp, wrp = os.path.join(run, "pmc_rd", "bench_counter_collection.csv"), os.path.join(run, "pmc_wr", "bench_counter_collection.csv")
    rd = {c: agg(rdp, "TCC_EA0_RDREQ%s_sum" % c) for c in ("", "_32B", "_64B", "_128B")}
    wr = {c: agg(wrp, "TCC_EA0_WRREQ%s_sum" % c) for c in ("", "_64B")}

    # k_huf_decode in the bench's kernel table is the whole a5 step (the probe brackets the stream-parallel launches and the serial
    # ones): its traffic record is the sum of both kernels; the table below also keeps k_huf_decode_par on a row of its own
    def fold(d):
        if "k_huf_decode_par" in d:
            t, p = d["k_huf_decode"], d["k_huf_decode_par"]
            t[1] += p[1]; t[2] += p[2]; t[0] = max(t[0], 1)
    for d in [fetch, write] + list(rd.values()) + list(wr.values()):
        fold(d)

    def req_bytes(k):
        if k not in rd[""] and k not in wr[""]:
            return None, None
        a, b64, c = rd["_32B"].get(k, [0, 0, 0])[1], rd["_64B"].get(k, [0, 0, 0])[1], rd["_128B"].get(k, [0, 0, 0])[1]
        tot = rd[""].get(k, [0, 0, 0])[1]
        other = max(tot - a - b64 - c, 0.0)                      # requests not in a size class are counted as 64 B
        rb = 32 * a + 64 * (b64 + other) + 128 * c
        w64, wt = wr["_64B"].get(k, [0, 0, 0])[1], wr[""].get(k, [0, 0, 0])[1]
        wb = 64 * w64 + 32 * max(wt - w64, 0.0)
        return rb, wb
    # passes over the data per kernel in the PMC run = launches / launches-per-pass
    lines = ["# %s: HBM traffic per kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" % tag, "",
             "PMC run: `python bench.py --steps 2 --warmup 1 --blocks %d --no-cpu-baseline` (3 passes over %d blocks)." % (nblocks, nblocks), ""]
    passes = 3
    hist_known = 32768.0 * nblocks * passes
    corr = 1.0
    if "k_hist" in fetch:
        raw = fetch["k_hist"][1] * 1024
        corr = hist_known / raw
        lines.append("Calibration on k_hist (reads every source byte exactly once): FETCH_SIZE*1024 = %.1f MB vs %.1f MB known "
                     "-> correction x%.3f (MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 on gfx950)." % (raw / 1e6, hist_known / 1e6, corr))
    corr_used = 2.0 if 1.8 < corr < 2.2 else 1.0
    lines += ["FETCH correction applied: x%.1f.  WRITE_SIZE calibrated on k_probagen (exact)." % corr_used, "",
              "| kernel | launches | fetch KiB (raw) | write KiB | HBM bytes / block (corrected) | read / write bytes per block by request size |", "|---|---|---|---|---|---|"]
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_") or k == "k_probagen":
            continue
        f = fetch.get(k, [0, 0, 0]); w = write.get(k, [0, 0, 0])
        per_block = (f[1] * 1024 * corr_used + w[1] * 1024) / (nblocks * passes)
        rb, wb = req_bytes(k)
        extra = "%.0f / %.0f" % (rb / (nblocks * passes), wb / (nblocks * passes)) if rb is not None else "-"
        lines.append("| %s | %d | %.0f | %.0f | %.0f | %s |" % (k, f[0], f[1], w[1], per_block, extra))
        rec = {"kernel": k, "hbm_bytes_per_block": round(per_block, 1), "fetch_correction": corr_used,
               "fetch_KiB_raw_total": f[1], "write_KiB_total": w[1], "blocks": nblocks, "passes": passes, "source": tag}
        if rb is not None:
            rec["request_size_bytes_per_block"] = {"read": round(rb / (nblocks * passes), 1), "write": round(wb / (nblocks * passes), 1)}
        json.dump(rec, open("profiles/traffic_%s%s.json" % (k, suffix), "w"))
    # kernel-trace run (scripts/profile.sh: bench.py --steps 5 --warmup 2 = 7 passes): durations per pass, the figure bench.py's
    # kernel_ms_per_step / roofline.avg_launch_ms report from HIP events.  A kernel that is launched once per decoder class shows
    # more calls than passes in rocprofv3's table -- the launches over classes without blocks return at once (a few microseconds)
    # and halve its "AverageNs"; the per-pass total is what agrees with the bench.
    trace_passes = 7
    stats = {}
    for r in csv.DictReader(open(os.path.join(run, "trace", "bench_kernel_stats.csv"))):
        k = r["Name"].split("(")[0]
        if k.startswith("void "):
            k = k[5:]
        k = k.split("<")[0]
        if k.startswith("k_") and k != "k_probagen":
            a = stats.setdefault(k, [0, 0.0, 0.0])
            a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"]); a[2] = max(a[2], float(r["MaxNs"]))
    lines += ["", "Kernel trace (`%s_kernel_stats.csv`, %d passes): per-pass totals = what `bench.py` reports per step." % (tag, trace_passes), "",
              "| kernel (all template instances) | calls | calls per pass | ms per pass | longest launch ms |", "|---|---|---|---|---|"]
    for k in sorted(stats, key=lambda k: -stats[k][1]):
        c, tot, mx = stats[k]
        lines.append("| %s | %d | %.1f | %.3f | %.3f |" % (k, c, c / trace_passes, tot / trace_passes / 1e6, mx / 1e6))
    open("profiles/%s_pmc.md" % tag, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    sq_summary(run, tag, nblocks, passes)


CUS_PER_SQ = 8.0     # SQ_BUSY_CYCLES sums one busy-cycle count per shader engine (8 XCDs x 4), SQ_LDS_IDX_ACTIVE the LDS-array cycles of all 256 CUs:
                     # checked against the kernel-trace durations (k_fse_decode: 6.4e8 busy cycles = 32 x kernel time x clock)
SQ_COUNTERS = ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
               "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY")


def sq_summary(run, tag, nblocks, passes):
    """profiles/<tag>_sq.md: the SQ counters of the two --pmc passes pmc_sq1 / pmc_sq2 per kernel, as ratios that can be read
    against DESIGN's statements about what bounds a kernel (MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
    quad-cycles; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES)."""
    vals = {}
    for sub in ("pmc_sq1", "pmc_sq2"):
        path = os.path.join(run, sub, "bench_counter_collection.csv")
        for c in SQ_COUNTERS:
            for k, a in agg(path, c).items():
                vals.setdefault(k, {})[c] = (a[1], a[0])
    if not vals:
        return
    lines = ["# %s: SQ counters per kernel (rocprofv3 --pmc, two passes of four counters; sums over the %d passes of %d blocks)" % (tag, passes, nblocks), "",
             "Reading: `valu` / `lds` = share of the wave cycles in which the wave had a VALU / LDS instruction in flight; `wait` = parked in "
             "s_waitcnt (memory or LDS results outstanding); `stall` = ready but not issued (pipe or dependency); `lds array busy` = LDS-array "
             "cycles (SQ_LDS_IDX_ACTIVE, all CUs) / (8 CUs per shader engine x SQ_BUSY_CYCLES) = share of the kernel's time a CU's LDS array is working; `conflict` = share of those cycles that are bank-conflict replays.", "",
             "| kernel | launches | wave cycles / block | valu | lds | wait | stall | lds array busy | conflict |", "|---|---|---|---|---|---|---|---|---|"]
    rec = {}
    for k in sorted(vals):
        if not k.startswith("k_") or k == "k_probagen":
            continue
        v = {c: vals[k].get(c, (0.0, 0))[0] for c in SQ_COUNTERS}
        wc = v["SQ_WAVE_CYCLES"] or 1.0
        row = {"wave_cycles_per_block": v["SQ_WAVE_CYCLES"] / (nblocks * passes), "valu": v["SQ_ACTIVE_INST_VALU"] / wc, "lds": v["SQ_ACTIVE_INST_LDS"] / wc,
               "wait": v["SQ_WAIT_ANY"] / wc, "stall": v["SQ_WAIT_INST_ANY"] / wc,
               "lds_array_busy": v["SQ_LDS_IDX_ACTIVE"] / (CUS_PER_SQ * (v["SQ_BUSY_CYCLES"] or 1.0)), "conflict": v["SQ_LDS_BANK_CONFLICT"] / (v["SQ_LDS_IDX_ACTIVE"] or 1.0)}
        rec[k] = {kk: round(vv, 4) for kk, vv in row.items()}
        rec[k]["raw"] = v
        lines.append("| %s | %d | %.0f | %.3f | %.3f | %.3f | %.3f | %.3f | %.3f |" % (k, vals[k].get("SQ_WAVE_CYCLES", (0, 0))[1], row["wave_cycles_per_block"], row["valu"], row["lds"],
                                                                                  row["wait"], row["stall"], row["lds_array_busy"], row["conflict"]))
    open("profiles/%s_sq.md" % tag, "w").write("\n".join(lines) + "\n")
    json.dump(rec, open("profiles/%s_sq.json" % tag, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
