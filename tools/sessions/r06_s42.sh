#!/bin/bash
# round 6, GPU session 42: the one-rank RCCL step in its three exchange forms on the final build (for tools/scale_model.py), next to the plain step
OUT=gpurun_out/r06s42
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
: > $OUT/forms.txt
for i in 1 2; do
  timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-psnr --sustained-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('plain      run $i: %.4f ms/step' % d['ms_per_step'])" >> $OUT/forms.txt
  for form in classic allreduce sharded; do
    PVD_DP_EXCHANGE=$form PVD_DP_FORCE=1 PVD_DP_PIPELINE=2 timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-psnr --sustained-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-10s run $i: %.4f ms/step   %s' % ('$form', d['ms_per_step'], d['config'].get('exchange','')[:110]))" >> $OUT/forms.txt
  done
done
cat $OUT/forms.txt
true
