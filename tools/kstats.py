"""Per-kernel (and per grid size) duration summary of a rocprofv3 --kernel-trace CSV: python tools/kstats.py trace.csv [substr ...]"""
import collections
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
pats = sys.argv[2:]
d = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if pats and not any(p in n for p in pats):
        continue
    d[(n[:60], r['Grid_Size_X'], r['Workgroup_Size_X'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print(f"{k[0]:60s} grid {k[1]:>8s} wg {k[2]:>5s} n={len(v):4d} min {v[0]/1e3:8.2f} med {v[len(v)//2]/1e3:8.2f} avg {sum(v)/len(v)/1e3:8.2f} max {v[-1]/1e3:8.2f} us")
