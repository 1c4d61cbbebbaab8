#!/bin/bash
# A/B of two builds of the library through PVD_HIP_LIB: ms/step of the default bench, alternating.   tools/ab_lib.sh <alt.so> [reps]
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/${TAG:-r03lib}; mkdir -p $OUT
ALT=$PWD/aaai2023-pvd_amd/$1; REPS=${2:-3}
for r in $(seq 1 $REPS); do for lib in "$ALT" ""; do
  PVD_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline ${ARGS:-} 2>>$OUT/err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('${lib:-in-tree build}'.split('/')[-1], 'ms/step %.4f' % d['ms_per_step'], 'loss %.4f' % d['config']['loss'])" | tee -a $OUT/ab_lib.txt
done; done
