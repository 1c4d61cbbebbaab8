#!/usr/bin/env python3
"""Which hardware queue each launch of one replayed step ran on (rocprofv3 kernel-trace CSV): the step between the last
two launches of `marker`, `skip` steps from the end.  python tools/step_queues.py <kernel_trace.csv> [marker] [skip]"""
import csv
import sys

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_adamw("
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 22
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
qcol = next((c for c in ("Queue_Id", "Queue_ID", "queue_id") if c in rows[0]), None)
scol = next((c for c in ("Stream_Id", "Stream_ID", "stream_id") if c in rows[0]), None)
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-skip - 1], idx[-skip]
t0 = int(rows[a]["End_Timestamp"])
print("columns:", ", ".join(rows[0].keys()))
print("%4s %9s %8s %6s %6s  %s" % ("#", "start us", "dur us", "queue", "stream", "kernel"))
end = t0
for i, r in enumerate(rows[a + 1:b + 1]):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    end = max(end, e)
    name = r["Kernel_Name"]
    name = name[name.find("k_"):][:40] if "k_" in name else name[:40]
    print("%4d %9.1f %8.1f %6s %6s  %s" % (i, (s - t0) / 1e3, (e - s) / 1e3, r.get(qcol, "?") if qcol else "?", r.get(scol, "?") if scol else "?", name))
print("step wall %.1f us; queues used in the whole trace: %s" % ((end - t0) / 1e3, sorted({r.get(qcol, "?") for r in rows}) if qcol else "?"))
