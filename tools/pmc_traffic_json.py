#!/usr/bin/env python3
"""Assemble profiles/r02_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
tools/pmc_teacher_fwd.py: mean KB per launch of k_hash_fwd_fused, the sample count, and the hash of the kernel sources
(bench.py quotes the record only for the build it was taken on).
  pmc_traffic_json.py <fetch_counter_collection.csv> <write_counter_collection.csv> <samples_per_launch> > r02_pmc_traffic.json"""
import csv
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import kernel_source_sha16  # noqa: E402


def mean_of(path, counter, kernel="k_hash_fwd_fused"):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)


fetch, n = mean_of(sys.argv[1], "FETCH_SIZE")
write, _ = mean_of(sys.argv[2], "WRITE_SIZE")
print(json.dumps({"kernel": "k_hash_fwd_fused", "fetch_kb": fetch, "write_kb": write, "samples_per_launch": int(sys.argv[3]), "launches": n,
                  "source_sha16": kernel_source_sha16(),
                  "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python tools/pmc_teacher_fwd.py; KB per "
                          "launch, raw (FETCH_SIZE counts the L2's 128-byte fabric requests at 64 B for wide streams; these are gathers: uncorrected)"},
                 indent=1))
