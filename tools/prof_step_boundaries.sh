#!/bin/bash
# The kernels either side of consecutive step boundaries inside one replayed 5-step graph (queues, gaps):
#   bash tools/prof_step_boundaries.sh [out]   (on the GPU box; writes gpurun_out/<out>/step_boundaries.txt)
OUT=${1:-r04k}
mkdir -p "$GRAFT_REPO_ROOT/gpurun_out/$OUT"
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/prof_sb && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_sb -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 100 --warmup 20 --no-cpu-baseline --no-psnr --sustained-steps 0 > /tmp/prof_sb.log 2>&1)
T=$(find /tmp/prof_sb -name "*kernel_trace.csv" | head -1)
for s in 60 59 58 57 56 55 54; do
  python "$GRAFT_REPO_ROOT/tools/step_queues.py" "$T" "k_adamw(" $s | grep -v "^columns"
done > "$GRAFT_REPO_ROOT/gpurun_out/$OUT/step_boundaries.txt" 2>&1
python "$GRAFT_REPO_ROOT/tools/step_walls.py" "$T" > "$GRAFT_REPO_ROOT/gpurun_out/$OUT/step_walls.txt" 2>&1
tail -3 /tmp/prof_sb.log | cut -c1-300
