"""Per-kernel MFMA utilisation from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` counter_collection.csv:
util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)   (MI355X_MICROARCH.md, counters section).
usage: mfma_summarize.py <counter_collection.csv> [kernel-name substring ...]"""
import collections
import csv
import json
import sys

want = sys.argv[2:] or ["rowkey_fwd", "dense_fwd", "dense_bwd"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if any(w in n for w in want):
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for n, c in acc.items():
    skip = 1 if len(next(iter(c.values()))) > 1 else 0            # first launch = warm-up
    avg = {k: sum(v[skip:]) / len(v[skip:]) for k, v in c.items()}
    e = {"launches": len(next(iter(c.values()))), "counters_avg_per_launch": avg}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and avg.get("GRBM_GUI_ACTIVE"):
        e["mfma_util"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * avg["GRBM_GUI_ACTIVE"] / 8)
    out[n] = e
print(json.dumps(out, indent=1))
