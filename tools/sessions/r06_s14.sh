#!/bin/bash
# round 6, GPU session 14: the same side-by-side under rocprofv3 --kernel-trace --stats: the kernels' own durations (the host-side call
# overhead of either binding out of the picture)
OUT=gpurun_out/r06s14
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
(cd /tmp && rm -rf /tmp/prof_ref && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ref -o b -- python "$GRAFT_REPO_ROOT/tools/bench_vs_reference_kernels.py" > /tmp/prof_ref.log 2>&1)
S=$(find /tmp/prof_ref -name "*kernel_stats.csv" | head -1)
python - "$S" <<'PY' | tee $OUT/vs_reference_kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ("kernel_near_far", "k_near_far", "kernel_march_rays_train", "k_march_count_wave", "k_march_write_records", "k_march_scan", "kernel_composite_rays_train_forward",
        "k_composite_fwd_wave", "kernel_composite_rays_train_backward", "k_composite_bwd_wave", "kernel_sh", "k_sh_fwd", "kernel_march_rays<", "kernel_march_rays(", "k_march_rays(")
print("%-100s %7s %10s" % ("kernel (rocprofv3 --kernel-trace --stats of tools/bench_vs_reference_kernels.py)", "calls", "avg us"))
for r in rows:
    n = r["Name"]
    if any(k in n for k in keep) or "march" in n or "composite" in n or "kernel_sh" in n or "near_far" in n:
        print("%-100s %7s %10.1f" % (n[:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
true
