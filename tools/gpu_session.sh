#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r02mlp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_m -o m -- python "$GRAFT_REPO_ROOT/tools/bench_mlp_to_tensors.py" > /tmp/prof_m.log 2>&1)
python tools/step_timeline.py $(find /tmp/prof_m -name "*kernel_trace.csv" | head -1) "k_adamw(" 3 > gpurun_out/r02mlp/timeline2.txt
cut -c1-130 gpurun_out/r02mlp/timeline2.txt
