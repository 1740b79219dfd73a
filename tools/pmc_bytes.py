"""Average of one PMC counter over the launches of one kernel in a rocprofv3 counter_collection.csv (first launch skipped).
usage: pmc_bytes.py <csv> <kernel substring> <counter>"""
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[3]]
print(sum(v[1:]) / len(v[1:]), len(v))
