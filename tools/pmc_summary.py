#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter-collection CSV per kernel: mean counter value per dispatch for every
libpvd_hip kernel (and the top torch kernels).  Usage: pmc_summary.py <counter_collection.csv> [...] > summary.csv"""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name") or row.get("Kernel Name")
            c, v = row["Counter_Name"], float(row["Counter_Value"])
            a = acc[k][c]
            a[0] += v
            a[1] += 1
import re


def short(k):
    """pvd kernels by their bare name (mangled or demangled, template arguments kept short); others truncated."""
    m = re.search(r"(k_[a-z0-9_]+)", k)
    if "pvd" in k and m:
        t = re.search(r"k_[a-z0-9_]+(<[^>]{0,24}>|I[A-Za-z0-9_]{0,16}E)?", k)
        return "pvd::" + (t.group(0) if t else m.group(1))
    return k[:80]


w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
for k in sorted(acc, key=lambda k: (not ("pvd" in k), k)):
    for c, (s, n) in sorted(acc[k].items()):
        w.writerow([short(k), c, n, s / n])
