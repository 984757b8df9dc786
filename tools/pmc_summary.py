#!/usr/bin/env python3
"""Per-kernel HBM traffic from the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs -- they do not fit one
TCC pass on gfx950). Units as rocprofv3 reports them (KB on gfx94x formulas); the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE tallies 128-B requests at 64 B -> x2 for wide coalesced reads) is applied
in the `fetch_bytes_corrected` column. Usage: pmc_summary.py <outdir> <workload>"""
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satdump_amd import build as sd_build  # noqa: E402


def collect(outdir, counter, wl):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"{outdir}/pmc_{counter}_{wl}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"]
            if "sdhip::" not in name:
                continue
            short = name.split("(")[0].replace("void ", "").replace("sdhip::", "")
            a = acc[short]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    outdir, wl = sys.argv[1], sys.argv[2]
    fe = collect(outdir, "FETCH_SIZE", wl)
    wr = collect(outdir, "WRITE_SIZE", wl)
    print("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1; counter unit = KB")
    print(f"# source_hash: {sd_build.source_hash()}")
    print("kernel,dispatches,fetch_KB_per_dispatch,fetch_bytes_corrected_x2,write_KB_per_dispatch,traffic_bytes_per_dispatch")
    for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, [0, 0])[1] + wr.get(k, [0, 0])[1])):
        nf, f = fe.get(k, [0, 0.0])
        nw, w = wr.get(k, [0, 0.0])
        fpd = f / nf if nf else 0.0
        wpd = w / nw if nw else 0.0
        print(f"{k},{max(nf, nw)},{fpd:.1f},{2 * fpd * 1024:.0f},{wpd:.1f},{(2 * fpd + wpd) * 1024:.0f}")


if __name__ == "__main__":
    main()
