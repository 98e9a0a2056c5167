"""development aid: fills DESIGN.md section 6's generated parts (@@...@@ markers, or the text between the BEGIN/END comments of an earlier fill) from
profiles/r06_bench_detail.json and profiles/r06_{fse,huf}_sq.md:  python scripts/design_fill.py r06"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
d = json.load(open(os.path.join(ROOT, "profiles", "%s_bench_detail.json" % tag)))

def sq_rows(path, wanted):
    out = {}
    for ln in open(path):
        c = [x.strip() for x in ln.strip().strip("|").split("|")]
        if len(c) == 9 and c[0] in wanted:
            out[c[0]] = c
    return out
sq = sq_rows(os.path.join(ROOT, "profiles", "%s_fse_sq.md" % tag), ("k_fse_decode", "k_fse_encode_wave", "k_hist", "k_fse_cprep", "k_fse_dbuild"))
sq.update(sq_rows(os.path.join(ROOT, "profiles", "%s_huf_sq.md" % tag), ("k_huf_decode_par", "k_huf_encode")))
pc = lambda v: "%.0f %%" % (100 * float(v))
lines = ["| kernel | wave cycles with a VALU instruction in flight | … an LDS instruction | parked in `s_waitcnt` | ready, not issued | LDS array busy | of which bank-conflict replays |", "|---|---|---|---|---|---|---|"]
for k in ("k_fse_decode", "k_fse_encode_wave", "k_hist", "k_huf_decode_par", "k_huf_encode"):
    c = sq[k]
    lines.append("| `%s` | %s | %s | %s | %s | %s | %s |" % (k, pc(c[3]), pc(c[4]), pc(c[5]), pc(c[6]), pc(c[7]), pc(c[8])))
a, b = sq["k_fse_cprep"], sq["k_fse_dbuild"]
lines.append("| `k_fse_cprep` / `k_fse_dbuild` | %s / %s | %s / %s | %s / %s | %s / %s | %s / %s | %s / %s |" % tuple(x for i in range(3, 9) for x in (pc(a[i]), pc(b[i]))))
sqtable = "\n".join(lines)

num = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "design_table.py"), os.path.join(ROOT, "profiles", "%s_bench_detail.json" % tag)], stdout=subprocess.PIPE, check=True).stdout.decode()
numtable = num.split("\n\nroofline:")[0].rstrip()

r, re_, s = d["roofline"], d["roofline_encode"], d["roofline"].get("secondary") or {}
w = s.get("where_the_rest_goes") or {}
nb = d["config"]["blocks_per_gpu"]
dominant = ("Dominant kernel `%s`: %.1f KB × %dk ÷ %.2f ms = %.0f GB/s = **%.1f %% of the HBM peak** (rounds 2 / 3 / 4 / 5: 4.8 / 5.5 / 6.5 / 6.5 %%); HBM traffic %.1f KB per block = "
            "%.2f × algorithmic (`profiles/%s_fse_pmc.md`).  The bound that holds is LDS capacity × chain latency: `roofline.secondary` = %d blocks per CU × 4 symbols per %.0f cycles "
            "(instrumented kernel; %.0f inside the phase) × %.2f GHz (measured clock) → model %.0f GB/s of output, achieved %.0f GB/s, achieved / model %.2f; what separates the two: "
            "%.1f %% idle workgroup slots (11.84 rounds of workgroups take 12), %.1f %% set-up, %.1f %% literal tails, %.1f %% finishing rounds, %.1f %% of decoder-wave waiting for the service waves "
            "(§4.3a; `EXPERIMENTS.md` §1 for what a per-slot refill could and could not recover).  The encoder's record (`roofline_encode`): `%s` %.2f ms per launch = %.0f GB/s = %.1f %% of the peak, "
            "traffic %.1f KB per block." % (
                r["kernel"], r["algorithmic_bytes_per_block"] / 1e3, nb // 1000, r["avg_launch_ms"], r["achieved"], 100 * r["frac"], r["traffic"] / nb / 1e3, r["traffic"] / nb / r["algorithmic_bytes_per_block"], tag,
                s.get("resident_blocks_per_cu", 33), s.get("cycles_per_iteration", 0), s.get("cycles_per_iteration_inside_the_phase", 0), s.get("clock_GHz", 0), s.get("model_GBps", 0), s.get("achieved_GBps", 0), s.get("frac", 0),
                100 * (1 - w.get("workgroup_slots_occupied", 1)), 100 * w.get("workgroup_time_setup", 0), 100 * w.get("workgroup_time_literal_tail", 0), 100 * w.get("workgroup_time_in_finishing_phases", 0),
                100 * w.get("decoder_wave_wait_frac", 0), re_["kernel"], re_["avg_launch_ms"], re_["achieved"], 100 * re_["frac"], re_["traffic"] / nb / 1e3))

p = os.path.join(ROOT, "DESIGN.md")
t = open(p).read()
for name, body in (("SQTABLE", sqtable), ("NUMTABLE", numtable), ("DOMINANT", dominant)):
    block = "<!-- BEGIN %s (scripts/design_fill.py) -->\n%s\n<!-- END %s -->" % (name, body, name)
    if "@@%s@@" % name in t:
        t = t.replace("@@%s@@" % name, block)
    else:
        t = re.sub(r"<!-- BEGIN %s \(scripts/design_fill.py\) -->.*?<!-- END %s -->" % (name, name), lambda m: block, t, flags=re.S)
open(p, "w").write(t)
print("filled", p)
