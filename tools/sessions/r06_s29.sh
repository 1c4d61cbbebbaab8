#!/bin/bash
# round 6, GPU session 29: where the teacher block forks the next step's march: at the start of the step vs behind the forward (A/B), timeline
OUT=gpurun_out/r06s29
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
: > $OUT/ab.txt
for i in 1 2 3; do
  for v in start forward; do
    PVD_TEACHER_FORK=$v timeout 300 python bench.py --workload teacher --steps 256 --warmup 320 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fork $v run $i: %.4f ms/step' % d['ms_per_step'])" | tee -a $OUT/ab.txt
  done
done
(cd /tmp && rm -rf /tmp/prof_t && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o b -- python "$GRAFT_REPO_ROOT/bench.py" --workload teacher --steps 64 --warmup 64 --no-cpu-baseline > /tmp/prof_t.log 2>&1)
T=$(find /tmp/prof_t -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $T "k_adamw(" 100 > $OUT/teacher_timeline.txt 2>&1; tail -24 $OUT/teacher_timeline.txt | cut -c1-150
timeout 900 python -m pytest tests/test_hip_workloads.py tests/test_hip_graph.py tests/test_hip_bench_line.py -q -x 2>&1 | tail -3
true
