#!/usr/bin/env python3
"""profiles/r04_in_step.json: the roofline kernel's duration INSIDE the replayed step, from the launch populations of a
rocprofv3 kernel trace of the driver's bench command (tools/kernel_populations.py), stamped with the hash of the kernel's
sources so that bench.py quotes it only for the build it was taken on.
  in_step_record.py <kernel_populations.txt> <bench_profiled_line.json> > profiles/r04_in_step.json"""
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import kernel_source_sha16  # noqa: E402

d = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1])
kernel = d["roofline"]["kernel"].split(" ")[0]
for l in open(sys.argv[1]):
    m = re.match(r"(\S+) sharing the chip.*?n=\s*(\d+)\s+mean ([\d.]+) us\s+median ([\d.]+)", l)
    if m and m.group(1) == kernel:
        print(json.dumps({"kernel": kernel, "launches": int(m.group(2)), "mean_us": float(m.group(3)), "median_us": float(m.group(4)),
                          "samples_per_launch": d["config"]["padded_rows_per_step"], "source_sha16": kernel_source_sha16(),
                          "command": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline",
                          "ms_per_step_of_that_run": d["ms_per_step"]}, indent=1))
        break
