"""development aid: the round's measurement table for DESIGN.md section 6 from a bench_detail.json (python scripts/design_table.py profiles/r06_bench_detail.json)"""
import json, sys
d = json.load(open(sys.argv[1]))
c = d["configs"]
k = d["kernel_ms_per_step"]
def ms(x, names): return ", ".join("%s %.2f" % (n.replace("k_", "").replace("fse_", "").replace("huf_", ""), x[n]) for n in names if n in x)
rows = []
rows.append("| **headline: P14 FSE, tableLog 11, decode maxLog 12** | **%s** (%.2f ms) | %.0f / %.0f | %s |" % (format(round(d["value"]), ","), d["ms_per_step"], d["encode_GBps"], d["decode_GBps"],
            "hist %.2f, cprep %.2f, encode %.2f, dparse+dbuild %.2f, **decode %.2f**" % (k["k_hist"], k["k_fse_cprep"], k["k_fse_encode_wave"], k["k_fse_dprep"], k["k_fse_decode"])))
x = c["cfg3_p80_fse"]; kk = x["kernel_ms_per_step"]
rows.append("| cfg 3: P80 FSE | %s (%.2f ms) | %.0f / %.0f | encode %.2f, decode %.2f |" % (format(round(x["value"]), ","), x["ms_per_step"], x["fse_encode_GBps"], x["fse_decode_GBps"], kk["k_fse_encode_wave"], kk["k_fse_decode"]))
x = c["cfg4_p14_huf"]; kk = x["kernel_ms_per_step"]
rows.append("| cfg 4: P14 Huff0 | %s (%.2f ms) | %.0f / %.0f | hist %.2f, cprep %.2f, encode %.2f, dprep %.2f, **decode %.2f** |" % (format(round(x["value"]), ","), x["ms_per_step"], x["huf_encode_GBps"], x["huf_decode_GBps"],
            kk["k_hist"], kk["k_huf_cprep"], kk["k_huf_encode"], kk["k_huf_dprep"], kk["k_huf_decode"]))
for key, name in (("fse_p14", "using tables, FSE P14 (`FSE_compress_usingCTable` + `FSE_decompress_usingDTable` over a batch, `maxTableLog` 11)"), ("fse_p80", "using tables, FSE P80"), ("huf_p14", "using tables, Huff0 P14")):
    x = c["using_tables"][key]; kk = x["kernel_ms_per_step"]
    enc = kk.get("k_fse_encode_wave", kk.get("k_huf_encode")); dec = kk.get("k_fse_decode", kk.get("k_huf_decode"))
    rows.append("| %s | %s (%.2f ms) | %.0f / %.0f | encode %.2f, decode %.2f |" % (name, format(round(x["value"]), ","), x["ms_per_step"], x["encode_GBps"], x["decode_GBps"], enc, dec))
x = c["cfg5_mixed_1M"]; kk = x["kernel_ms_per_step"]
rows.append("| cfg 5 as named: 1M mixed blocks, FSE + Huff0, one GPU — all 1M blocks of both codecs against the reference | %s (%.1f ms) | FSE %.0f / %.0f, Huff0 %.0f / %.0f | FSE decode %.1f, FSE encode %.1f, Huff0 decode %.1f … |" % (
    format(round(x["value"]), ","), x["ms_per_step"], x["fse_encode_GBps"], x["fse_decode_GBps"], x["huf_encode_GBps"], x["huf_decode_GBps"], kk["k_fse_decode"], kk["k_fse_encode_wave"], kk["k_huf_decode"]))
x = c["cfg5_mixed_shard"]; kk = x["kernel_ms_per_step"]
rows.append("| cfg 5 shard: 125k mixed blocks | %s (%.1f ms) | FSE %.0f / %.0f, Huff0 %.0f / %.0f | FSE decode %.1f, encode %.1f, Huff0 decode %.1f |" % (
    format(round(x["value"]), ","), x["ms_per_step"], x["fse_encode_GBps"], x["fse_decode_GBps"], x["huf_encode_GBps"], x["huf_decode_GBps"], kk["k_fse_decode"], kk["k_fse_encode_wave"], kk["k_huf_decode"]))
x = c["fse_tl12"]; kk = x["kernel_ms_per_step"]
rows.append("| `fse -b` table log 12, P14 | %s (%.1f ms) | %.0f / %.0f | decode %.1f (8 KiB tables: 18 blocks per CU), encode %.1f |" % (format(round(x["value"]), ","), x["ms_per_step"], x["fse_encode_GBps"], x["fse_decode_GBps"], kk["k_fse_decode"], kk["k_fse_encode_wave"]))
x = c["huf_tl12"]; kk = x["kernel_ms_per_step"]
rows.append("| Huff0 tableLog 12, P02 | %s (%.2f ms) | %.0f / %.0f | cprep %.1f, encode %.1f, decode %.1f |" % (format(round(x["value"]), ","), x["ms_per_step"], x["huf_encode_GBps"], x["huf_decode_GBps"], kk["k_huf_cprep"], kk["k_huf_encode"], kk["k_huf_decode"]))
x = c["fse_u16"]
rows.append("| 16-bit symbols: 25k × 16,384 symbols, 287-symbol alphabet (frozen since round 5) | %s (%.2f ms) | %.0f / %.0f | – |" % (format(round(x["value"]), ","), x["ms_per_step"], x["encode_GBps"], x["decode_GBps"]))
r = c.get("ragged")
if r and "fse" in r:
    rows.append("| ragged batch: 20k P14 blocks of 12,000 … 32,768 bytes (uniform 20k × 32 KB beside it) | – | FSE %.0f / %.0f (uniform %.0f / %.0f), Huff0 %.0f / %.0f (%.0f / %.0f) | per byte vs uniform: FSE decode %.2f, Huff0 decode %.2f |" % (
        r["fse"]["ragged"]["encode_GBps"], r["fse"]["ragged"]["decode_GBps"], r["fse"]["uniform"]["encode_GBps"], r["fse"]["uniform"]["decode_GBps"],
        r["huf"]["ragged"]["encode_GBps"], r["huf"]["ragged"]["decode_GBps"], r["huf"]["uniform"]["encode_GBps"], r["huf"]["uniform"]["decode_GBps"],
        r["fse"]["ragged_over_uniform_per_byte"]["decode"], r["huf"]["ragged_over_uniform_per_byte"]["decode"]))
h = d["host_inclusive"]
rows.append("| host buffers (`host_inclusive`), fixed-stride slots | %s | %.1f / %.1f | %.1f / %.1f ms per direction for 100k blocks |" % (format(round(h["value"]), ","), h["encode_GBps"], h["decode_GBps"], h["encode_ms"], h["decode_ms"]))
p = h["packed"]
rows.append("| host buffers, packed results (`host_inclusive.packed`) | %s | %.1f / %.1f | %.1f / %.1f ms |" % (format(round(p["value"]), ","), p["encode_GBps"], p["decode_GBps"], p["encode_ms"], p["decode_ms"]))
b = d["cpu_baseline"]
rows.append("| reference, %d host threads (same box, %s) | %s | %.1f / %.1f | `cpu_baseline`, kind \"%s\"; one thread: %.0f |" % (b["cores"], b["cpu_model"], format(round(b["value"]), ","), b["encode_MiBps"] * 1.048576 / 1000, b["decode_MiBps"] * 1.048576 / 1000, b["kind"], b["single_thread_value"]))
print("| configuration | value (MiB/s round trip) | encode / decode GB/s | kernel ms per step |\n|---|---|---|---|")
print("\n".join(rows))
s = d["roofline"].get("secondary") or {}
print("\nroofline:", {k2: d["roofline"][k2] for k2 in ("kernel", "achieved", "frac", "traffic", "avg_launch_ms")}, "\nsecondary:", {k2: s.get(k2) for k2 in ("cycles_per_iteration", "cycles_per_iteration_inside_the_phase", "model_GBps", "achieved_GBps", "frac", "clock_GHz")}, s.get("where_the_rest_goes"))
print("roofline_encode:", {k2: d["roofline_encode"][k2] for k2 in ("kernel", "achieved", "frac", "traffic", "avg_launch_ms")})
