#!/usr/bin/env python3
"""Group a rocprofv3 kernel-trace CSV into consecutive runs of `n` dispatches of kernels matching `pattern`
and print the mean duration of the last `keep` of each run (for microbenchmarks that launch each
configuration warmup+iters times in a fixed order).
  python tools/trace_groups.py <kernel_trace.csv> <pattern> <n> <keep> [label ...]"""
import csv
import sys

path, pat, n, keep = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
labels = sys.argv[5:]
rows = [r for r in csv.DictReader(open(path)) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for g in range(len(rows) // n):
    grp = rows[g * n:(g + 1) * n][-keep:]
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in grp]
    lab = labels[g] if g < len(labels) else "group %d" % g
    print("%-34s mean %8.2f us  min %8.2f us" % (lab, sum(d) / len(d) / 1e3, min(d) / 1e3))
