#!/bin/bash
# round 6, GPU session 27: timeline of one teacher-training step (configs[1]) inside its 16-step graph
OUT=gpurun_out/r06s27
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
(cd /tmp && rm -rf /tmp/prof_t && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o b -- python "$GRAFT_REPO_ROOT/bench.py" --workload teacher --steps 64 --warmup 64 --no-cpu-baseline > /tmp/prof_t.log 2>&1)
T=$(find /tmp/prof_t -name "*kernel_trace.csv" | head -1)
grep '^{' /tmp/prof_t.log | tail -1 | cut -c1-300
python tools/step_timeline.py $T "k_adamw(" 100 > $OUT/teacher_timeline.txt 2>&1; tail -40 $OUT/teacher_timeline.txt
true
