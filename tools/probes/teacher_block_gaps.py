#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of `bench.py --workload teacher`: what happens between two 16-step graphs (the occupancy-grid update):
wall time from the last update kernel of a block to the first kernel of the next block, kernel-busy time inside it, idle time.
    python tools/probes/teacher_block_gaps.py <kernel_trace.csv>"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
adam = [i for i, r in enumerate(rows) if r[2].startswith("pvd::k_adamw(") or "k_adamw(" in r[2]]
# a block boundary: an update kernel followed (before the next update kernel) by occupancy kernels
out = []
for a, b in zip(adam[:-1], adam[1:]):
    between = rows[a + 1:b]
    if not any("k_occ" in r[2] for r in between):
        continue
    t0 = rows[a][1]
    first_step = next((r for r in between if "k_grid_fwd" in r[2] and r[0] > max(x[1] for x in between if "k_occ" in x[2])), None)
    if first_step is None:
        continue
    t1 = first_step[0]
    inside = [r for r in between if r[0] < t1]
    # union of busy intervals
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in sorted(inside):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    out.append(((t1 - t0) / 1e3, busy / 1e3, len(inside), sum(e - s for s, e, n in inside if "k_occ" in n) / 1e3))
print("%d block boundaries" % len(out))
for w, b, n, occ in out[-8:]:
    print("update window %8.1f us   kernels busy %8.1f us (%d launches; k_occ_* %6.1f us)   idle %8.1f us" % (w, b, n, occ, w - b))
if out:
    import statistics
    print("median window %.1f us, busy %.1f us, idle %.1f us -> per step (16): %.1f us of which idle %.1f" % (
        statistics.median(o[0] for o in out), statistics.median(o[1] for o in out), statistics.median(o[0] - o[1] for o in out),
        statistics.median(o[0] for o in out) / 16, statistics.median(o[0] - o[1] for o in out) / 16))
# the launches of the last window
last = None
for a, b in zip(adam[:-1], adam[1:]):
    between = rows[a + 1:b]
    if any("k_occ" in r[2] for r in between):
        last = (rows[a][1], between)
if last:
    t0, between = last
    print("launches of the last update window:")
    for s, e, n in between[:30]:
        print("  +%8.1f us  %8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n[:110]))
