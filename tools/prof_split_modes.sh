#!/bin/bash
# PVD_ADAMW_SPLIT=0 / late under rocprofv3 --kernel-trace on ONE box: populations of the kernels around the step boundary, step walls.
OUT=${1:-r04m}
mkdir -p "$GRAFT_REPO_ROOT/gpurun_out/$OUT"
export TMPDIR=/tmp
for mode in 0 late; do
  (cd /tmp && rm -rf /tmp/prof_sm && PVD_ADAMW_SPLIT=$mode timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_sm -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 100 --warmup 20 --no-cpu-baseline --no-psnr --sustained-steps 0 > /tmp/prof_sm.log 2>&1)
  T=$(find /tmp/prof_sm -name "*kernel_trace.csv" | head -1)
  echo "== PVD_ADAMW_SPLIT=$mode $(grep -o '"ms_per_step": [0-9.]*' /tmp/prof_sm.log | head -1)"
  for k in k_vm_fwd k_march_count_wave "k_adamw(" k_vm_bwd_split k_head_bwd k_head_fwd k_hash_fwd_fused; do python "$GRAFT_REPO_ROOT/tools/kernel_populations.py" $T "$k" | grep sharing | cut -c1-150; done
  python "$GRAFT_REPO_ROOT/tools/step_walls.py" $T k_vm_bwd_split 100 | tail -1
  python "$GRAFT_REPO_ROOT/tools/step_queues.py" $T k_vm_bwd_split 50 | grep -v columns | cut -c1-90
done > "$GRAFT_REPO_ROOT/gpurun_out/$OUT/split_modes.txt" 2>&1
