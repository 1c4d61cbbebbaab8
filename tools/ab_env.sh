#!/bin/bash
# A/B of one environment knob on the default bench: tools/ab_env.sh VAR a b [reps] ; alternating, ms/step.
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/${TAG:-r03env}; mkdir -p $OUT
V=$1; A=$2; B=$3; REPS=${4:-3}
for r in $(seq 1 $REPS); do for x in $A $B; do
  env $V=$x timeout 300 python bench.py --no-cpu-baseline ${ARGS:-} 2>>$OUT/err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$V=$x', 'ms/step %.4f' % d['ms_per_step'], 'loss %.4f psnr %.2f' % (d['config']['loss'], d['config']['psnr_student_vs_teacher_db']))" | tee -a $OUT/ab_env.txt
done; done
