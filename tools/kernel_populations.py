#!/usr/bin/env python3
"""Launches of one kernel in a rocprofv3 kernel-trace CSV of `bench.py`, split into the populations the run contains:
the launches that overlap another kernel in time (inside the replayed step the frozen teacher's forward runs on a forked
branch next to the student's scatter) and the launches that have the chip to themselves (the roofline launches right after
the timed region, teacher pre-training, warm-up).
  python tools/kernel_populations.py <kernel_trace.csv> [kernel substring]"""
import csv
import sys

path = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "k_hash_fwd_fused"
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
starts = [r[0] for r in rows]
import bisect
alone, shared = [], []
for i, (s, e, k) in enumerate(rows):
    if name not in k:
        continue
    overlap = 0
    j = i - 1
    while j >= 0 and j >= i - 8:  # a neighbour that started earlier and is still running
        if rows[j][1] > s:
            overlap += min(rows[j][1], e) - s
        j -= 1
    j = bisect.bisect_right(starts, s, lo=i + 1)
    for q in range(i + 1, min(len(rows), i + 9)):
        if rows[q][0] < e:
            overlap += min(rows[q][1], e) - rows[q][0]
    (shared if overlap > 0.2 * (e - s) else alone).append((e - s) / 1e3)
for label, v in (("alone on the chip", alone), ("sharing the chip (forked branch of the replayed step)", shared)):
    if v:
        v.sort()
        print("%s %-52s n=%4d  mean %.2f us  median %.2f  min %.2f  max %.2f" % (name, label, len(v), sum(v) / len(v), v[len(v) // 2], v[0], v[-1]))
