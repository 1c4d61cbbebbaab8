#!/bin/bash
# round 6, GPU session 30: kernel statistics of the other students' steps (anything that takes far longer than its work?)
OUT=gpurun_out/r06s30
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
run() {
  tag=$1; shift
  (cd /tmp && rm -rf /tmp/prof_$tag && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o b -- python "$GRAFT_REPO_ROOT/bench.py" "$@" --steps 100 --warmup 20 --no-cpu-baseline --no-psnr --sustained-steps 0 > /tmp/prof_$tag.log 2>&1)
  grep '^{' /tmp/prof_$tag.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag: %.4f ms/step' % d['ms_per_step'])"
  T=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
  python tools/step_timeline.py $T "k_adamw(" ${IDX:-260} 2>&1 | tail -40 | cut -c1-140 > $OUT/timeline_$tag.txt
  tail -3 $OUT/timeline_$tag.txt
}
run hash_student --student hash --teacher-pretrain 100
run tensors_student --student tensors --teacher-pretrain 100
run config3 --teacher mlp --student tensors --data-type llff --teacher-pretrain 0
run config4 --student hash --data-type tank --bound 2 --dt-gamma 0.00390625 --scene-scale 1.9 --teacher-pretrain 100
true
