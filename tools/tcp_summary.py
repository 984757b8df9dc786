#!/usr/bin/env python3
"""Per-kernel table of whatever counters one rocprofv3 --pmc pass collected (sum over the kernel's dispatches; duration from the kernel trace of the same pass).
usage: tcp_summary.py <dir of the pass>"""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float)); names = set()
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "sdhip::" not in n: continue
        s = n.split("(")[0].replace("void ", "").replace("sdhip::", "")
        acc[s][r["Counter_Name"]] += float(r["Counter_Value"]); names.add(r["Counter_Name"])
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "sdhip::" not in n: continue
        s = n.split("(")[0].replace("void ", "").replace("sdhip::", "")
        acc[s]["dur_us"] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
cols = ["dur_us"] + sorted(names)
print("kernel," + ",".join(cols))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["dur_us"])[:10]:
    print(k + "," + ",".join(f"{v[c]:.0f}" for c in cols))
