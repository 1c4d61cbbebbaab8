#!/usr/bin/env python3
"""Wall time of every step of a profiled bench run (rocprofv3 kernel-trace CSV): marker end to marker end, plus the gap from the
marker's end to the first launch of the next step that runs on the marker's own queue.
  python tools/step_walls.py <kernel_trace.csv> [marker] [last_n]"""
import csv
import sys

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_adamw("
last = int(sys.argv[3]) if len(sys.argv) > 3 else 160
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]][-last:]
walls, gaps = [], []
for a, b in zip(idx[:-1], idx[1:]):
    end_a = int(rows[a]["End_Timestamp"])
    walls.append((int(rows[b]["End_Timestamp"]) - end_a) / 1e3)
    q = rows[a].get("Queue_Id")
    nxt = next((r for r in rows[a + 1:b] if r.get("Queue_Id") == q), None)
    gaps.append((int(nxt["Start_Timestamp"]) - end_a) / 1e3 if nxt else float("nan"))
print("step walls (us):", " ".join("%.0f" % w for w in walls))
print("gap to the next launch on the marker's queue (us):", " ".join("%.1f" % g for g in gaps))
s = sorted(walls)
print("n %d  median %.1f  mean %.1f  min %.1f  max %.1f" % (len(s), s[len(s) // 2], sum(s) / len(s), s[0], s[-1]))
