#!/bin/bash
# ms/step of the replayed distillation step for the fork placements of the pipelined multi-step graph (PVD_PIPELINE_FORK):
# mid (between head backward and scatter; default so far), start (before the student's forward), graph (one fork per graph).
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/${TAG:-r03fork}; mkdir -p $OUT
for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20"; do
  for m in ${MODES:-mid start graph}; do
    for rep in 1 2; do
      PVD_PIPELINE_FORK=$m timeout 300 python bench.py $args --no-cpu-baseline 2>>$OUT/err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$m', '$args', 'ms/step %.4f' % d['ms_per_step'], '| hash alone %.1f us |' % d['roofline']['alone']['us_per_launch'], d['config']['launch'][16:110])" | tee -a $OUT/fork_modes.txt
    done
  done
done
