#!/bin/bash
# round 6, GPU session 45: the one-rank RCCL step (ray-DP recording, collectives in the graph) against the plain step for the other students
OUT=gpurun_out/r06s45
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
: > $OUT/forms.txt
run() {
  tag=$1; shift
  for dpf in 0 1; do
    PVD_DP_FORCE=$dpf PVD_DP_PIPELINE=2 timeout 300 python bench.py "$@" --steps 400 --warmup 40 --no-cpu-baseline --no-psnr --sustained-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-28s %s: %.4f ms/step   %s' % ('$tag', 'one-rank RCCL' if '$dpf' == '1' else 'plain        ', d['ms_per_step'], d['config'].get('exchange','')[:150]))" >> $OUT/forms.txt
  done
}
run "hash->tensors" --student tensors --teacher-pretrain 100
run "mlp->tensors llff (configs[3])" --teacher mlp --student tensors --data-type llff --teacher-pretrain 0
run "hash->hash tank (configs[4])" --student hash --data-type tank --bound 2 --dt-gamma 0.00390625 --scene-scale 1.9 --teacher-pretrain 100
cat $OUT/forms.txt
true
