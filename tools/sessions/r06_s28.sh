#!/bin/bash
# round 6, GPU session 28: the teacher-training step's launch diet (configs[1]): tests, bench line, timeline
OUT=gpurun_out/r06s28
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_hip_workloads.py tests/test_hip_fullsize.py tests/test_hip_graph.py tests/test_hip_fused_misc.py tests/test_hip_budget.py tests/test_hip_golden.py tests/test_hip_golden_step.py tests/test_hip_bench_line.py -q -x 2>&1 | tail -5 | tee $OUT/tests.log
for i in 1 2 3; do
timeout 300 python bench.py --workload teacher --steps 256 --warmup 320 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('teacher %.4f ms/step' % d['ms_per_step'])" | tee -a $OUT/bench.txt
done
timeout 300 python bench.py --student hash --no-cpu-baseline --no-psnr --teacher-pretrain 100 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hash student %.4f ms/step' % d['ms_per_step'])" | tee -a $OUT/bench.txt
(cd /tmp && rm -rf /tmp/prof_t && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o b -- python "$GRAFT_REPO_ROOT/bench.py" --workload teacher --steps 64 --warmup 64 --no-cpu-baseline > /tmp/prof_t.log 2>&1)
T=$(find /tmp/prof_t -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $T "k_adamw(" 100 > $OUT/teacher_timeline.txt 2>&1; tail -32 $OUT/teacher_timeline.txt | cut -c1-150
true
