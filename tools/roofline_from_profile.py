#!/usr/bin/env python3
"""Re-derive bench.py's `roofline` figures from the committed rocprofv3 evidence, without a GPU:

    python tools/roofline_from_profile.py [profiles/r06_kernel_populations.txt] [profiles/r06_bench_profiled_line.json] [profiles/r06_kernel_stats.csv]

  kernel_populations.txt   tools/kernel_populations.py over the kernel trace of `rocprofv3 --kernel-trace --stats -- python bench.py
                           <the driver's arguments>`: the launches of the roofline kernel split into those that have the chip to
                           themselves (bench.py's roofline launches after the timed region) and those inside the replayed step
                           (forked branch of the graph, sharing the chip with the student's scatter and update)
  bench_profiled_line.json the JSON line that same profiled run printed (samples per launch, bytes per sample, its own live figures)
  kernel_stats.csv         rocprofv3's per-kernel summary of the run (all launches of a kernel in one average)

Prints, for each population and for rocprofv3's all-launch average: microseconds per launch, GB/s of ALGORITHMIC bytes
(516 B per sample for the f16 lookup x the samples of one launch) and the fraction of the 8 TB/s HBM peak --
next to the figures the line itself carries, with the relative differences."""
import csv
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(REPO, "profiles")
pops = sys.argv[1] if len(sys.argv) > 1 else os.path.join(P, "r06_kernel_populations.txt")
line = sys.argv[2] if len(sys.argv) > 2 else os.path.join(P, "r06_bench_profiled_line.json")
stats = sys.argv[3] if len(sys.argv) > 3 else os.path.join(P, "r06_kernel_stats.csv")

d = json.loads([l for l in open(line).read().splitlines() if l.startswith("{")][-1])
r = d["roofline"]
kernel = r["kernel"].split(" ")[0]
rows, bps, peak = r["samples_per_launch"], r["bytes_per_sample"], r["peak"]
print("kernel %s: %d samples per launch x %d B = %.2f MB algorithmic per launch; peak %.0f GB/s" % (kernel, rows, bps, rows * bps / 1e6, peak))


def show(label, us, n=None, ref=None):
    gbs = rows * bps / (us * 1e-6) / 1e9
    s = "  %-46s %7.2f us%s -> %7.1f GB/s = %.4f of peak" % (label, us, "" if n is None else " (n=%d)" % n, gbs, gbs / peak)
    if ref is not None:
        s += "   | the line says %.4f (%+.1f %%)" % (ref, 100.0 * (gbs / peak - ref) / ref)
    print(s)
    return gbs / peak


for l in open(pops):
    m = re.match(r"(\S+) (.*?)\s+n=\s*(\d+)\s+mean ([\d.]+) us\s+median ([\d.]+)", l)
    if not m or m.group(1) != kernel:
        continue
    alone = m.group(2).startswith("alone")
    ref = (r.get("alone") or {}).get("frac") if alone else (r.get("in_step") or {}).get("frac")
    show(("alone on the chip" if alone else "inside the replayed step") + ", median", float(m.group(5)), int(m.group(3)), ref)
    show(("alone on the chip" if alone else "inside the replayed step") + ", mean", float(m.group(4)), int(m.group(3)))
if os.path.exists(stats):
    for row in csv.DictReader(open(stats)):
        if kernel in row.get("Name", ""):
            show("rocprofv3 --stats average over ALL launches", float(row["AverageNs"]) / 1e3, int(row["Calls"]))
print("the line's headline: frac %.4f (%s, %.2f us per launch, %d launches)" % (r["frac"], r.get("where", "?"), r["us_per_launch"], r["launches"]))
