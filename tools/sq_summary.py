#!/usr/bin/env python3
"""Per-kernel SQ counter table from tools/gpu_visit.sh's sq stage (sum over dispatches of the kernel; duration from the kernel trace)."""
import csv, glob, sys
from collections import defaultdict
outdir, wl = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float)); ndis = defaultdict(set)
for tag in ("sq", "sq2"):
    for f in glob.glob(f"{outdir}/{tag}_{wl}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "sdhip::" not in n: continue
            s = n.split("(")[0].replace("void ", "").replace("sdhip::", "")
            acc[s][r["Counter_Name"]] += float(r["Counter_Value"])
            if tag == "sq": ndis[s].add(r["Dispatch_Id"])
    for f in glob.glob(f"{outdir}/{tag}_{wl}/**/*kernel_trace.csv", recursive=True):
        if tag != "sq": continue
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "sdhip::" not in n: continue
            s = n.split("(")[0].replace("void ", "").replace("sdhip::", "")
            acc[s]["dur_us"] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
cols = ["dur_us", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY",
        "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INST_CYCLES_VMEM", "GRBM_GUI_ACTIVE"]
print("kernel,dispatches," + ",".join(cols) + ",valu_Ginst_per_s(pmc-run)")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["dur_us"]):
    rate = v["SQ_INSTS_VALU"] / (v["dur_us"] * 1e-6) / 1e9 if v["dur_us"] else 0
    print(f"{k},{len(ndis[k])}," + ",".join(f"{v[c]:.0f}" for c in cols) + f",{rate:.1f}")
