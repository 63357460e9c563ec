#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files for one kernel."""
import csv, glob, sys, json
d = sys.argv[1]; kernel = sys.argv[2] if len(sys.argv) > 2 else "render_kernel"
tot = {}; disp = {}
for f in glob.glob(d + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if kernel not in r["Kernel_Name"]: continue
        k = r["Counter_Name"]; tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"]); disp.setdefault(k, set()).add(r["Dispatch_Id"])
out = {k: tot[k] / len(disp[k]) for k in sorted(tot)}   # per launch
print(json.dumps(out, indent=1))
