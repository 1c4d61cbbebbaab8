#!/bin/bash
TAG=${TAG:-r02final4}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench_line.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline > /tmp/prof_b.log 2>&1)
cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
python tools/step_timeline.py $(find /tmp/prof_b -name "*kernel_trace.csv" | head -1) "k_adamw(" 22 > $OUT/step_timeline.txt; tail -4 $OUT/step_timeline.txt
timeout 300 python bench.py --student tensors --no-cpu-baseline --teacher-pretrain 100 > $OUT/bench_tensors_student.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_tensors_student.json
timeout 300 python tools/bench_mlp_to_tensors.py 2>&1 | grep -v amdgpu > $OUT/bench_mlp_to_tensors.txt; cat $OUT/bench_mlp_to_tensors.txt
