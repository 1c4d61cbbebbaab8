#!/usr/bin/env python3
"""One steady-state step of bench.py from a rocprofv3 kernel-trace CSV: the kernels between the last two
launches of `marker` (default: the flat AdamW kernel), in launch order, with durations and gaps.
  python tools/step_timeline.py <kernel_trace.csv> [marker] [skip_from_end]"""
import csv
import sys

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_adamw("
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 45  # steps from the end to look at (roofline re-launches follow the timed region)
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-skip - 1], idx[-skip]
step = rows[a + 1:b + 1]
t0 = int(rows[a]["End_Timestamp"])
busy = 0
print("%4s %9s %8s %8s  %s" % ("#", "start us", "dur us", "gap us", "kernel"))
prev_end = t0
for i, r in enumerate(step):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    print("%4d %9.1f %8.1f %8.1f  %s" % (i, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:110]))
    prev_end = max(prev_end, e)
print("step wall %.1f us, kernels busy %.1f us, %d launches" % ((prev_end - t0) / 1e3, busy / 1e3, len(step)))
