#!/usr/bin/env python3
"""Condense a rocprofv3 result (rocpd sqlite .db, or a *_kernel_stats.csv) into the per-kernel summary that is
committed under profiles/: name, calls, total us, average us, % of GPU kernel time. torch's input-synthesis kernels
(at::native::*) are folded into one line -- they run before the timed region of bench.py."""
import csv
import sqlite3
import sys


def rows_from_db(path):
    c = sqlite3.connect(path)
    return [(r[0], int(r[1]), float(r[2]), float(r[3])) for r in c.execute("select name,total_calls,total_duration,average from top_kernels")]


def rows_from_csv(path):
    out = []
    for r in csv.DictReader(open(path)):
        out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
    return out


def main():
    path = sys.argv[1]
    rows = rows_from_db(path) if path.endswith(".db") else rows_from_csv(path)
    tot = sum(r[2] for r in rows)
    ours, other_calls, other_us = [], 0, 0.0
    for name, calls, total, avg in rows:
        if "sdhip::" in name:
            short = name.split("(")[0].replace("void ", "").replace("sdhip::", "")
            ours.append((short, calls, total, avg))
        else:
            other_calls += calls
            other_us += total
    ours.sort(key=lambda r: -r[2])
    print(f"# rocprofv3 --kernel-trace --stats summary of: {' '.join(sys.argv[2:]) or path}")
    print(f"# total GPU kernel time {tot / 1e3:.3f} ms; sdhip kernels {sum(r[2] for r in ours) / 1e3:.3f} ms")
    print("kernel,calls,total_us,avg_us,pct_of_all_kernel_time")
    for short, calls, total, avg in ours:
        print(f"{short},{calls},{total:.1f},{avg:.1f},{100 * total / tot:.2f}")
    print(f"(torch input synthesis + copies: at::native::* / rocclr),{other_calls},{other_us:.1f},,{100 * other_us / tot:.2f}")


if __name__ == "__main__":
    main()
