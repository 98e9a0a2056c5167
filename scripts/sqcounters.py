"""development aid: per-kernel SQ counter ratios from a `rocprofv3 --pmc SQ_...` csv (fractions of SQ_WAVE_CYCLES)"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
names = sorted({n for d in agg.values() for n in d} - {"SQ_WAVE_CYCLES"})
for k in sorted(agg):
    if not k.startswith("k_"):
        continue
    d = agg[k]
    wc = d["SQ_WAVE_CYCLES"] or 1
    print("%-20s" % k, " ".join("%s=%.3f" % (n.replace("SQ_", ""), d[n] / wc) for n in names), "wavecyc=%.3g" % wc)
