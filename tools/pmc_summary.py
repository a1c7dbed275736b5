"""Sums the counters of a rocprofv3 --pmc CSV per kernel name.   usage: python tools/pmc_summary.py <counter_collection.csv> [name filter]"""
import csv, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); calls = defaultdict(set)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0]
    if flt and flt not in name: continue
    tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[name].add(r["Dispatch_Id"])
for name, c in tot.items():
    print(name, "dispatches", len(calls[name]))
    for k, v in sorted(c.items()):
        print("   %-28s %.4g" % (k, v))
