#!/bin/bash
# One replayed step's launches (queues, starts, durations) under a given PVD_PIPELINE_FORK mode:
#   bash tools/prof_fork_mode.sh <mode> [out]     (GPU box; writes gpurun_out/<out>/fork_<mode>.txt)
MODE=${1:-optimizer}
OUT=${2:-r04l}
mkdir -p "$GRAFT_REPO_ROOT/gpurun_out/$OUT"
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/prof_fm && PVD_PIPELINE_FORK=$MODE timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_fm -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 100 --warmup 20 --no-cpu-baseline --no-psnr --sustained-steps 0 > /tmp/prof_fm.log 2>&1)
T=$(find /tmp/prof_fm -name "*kernel_trace.csv" | head -1)
{ grep -o '"ms_per_step": [0-9.]*' /tmp/prof_fm.log; for s in 57 56; do python "$GRAFT_REPO_ROOT/tools/step_queues.py" "$T" "k_adamw(" $s | grep -v "^columns"; done; python "$GRAFT_REPO_ROOT/tools/step_walls.py" "$T" "k_adamw(" 100 | tail -1; } > "$GRAFT_REPO_ROOT/gpurun_out/$OUT/fork_$MODE.txt" 2>&1
