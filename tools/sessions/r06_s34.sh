#!/bin/bash
# round 6, GPU session 34: HIP runtime knobs round 4 did not cover, against the replayed step (bench.py --steps 400 --warmup 40), two rounds.
OUT=gpurun_out/r06s34
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
: > $OUT/runtime_knobs.txt
run() {
  env "$@" timeout 120 python bench.py --steps 400 --warmup 40 --no-psnr --no-cpu-baseline --sustained-steps 0 2>/dev/null | python -c "
import sys, json
l = [x for x in sys.stdin if x.startswith('{')]
print('%.4f' % json.loads(l[0])['ms_per_step'] if l else 'FAILED')"
}
for round in 1 2; do
  for s in "X=0" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "DEBUG_HIP_KERNARG_COPY_OPT=0" "DEBUG_HIP_KERNARG_COPY_OPT=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=1000" "DEBUG_HIP_DYNAMIC_QUEUES=0" "DEBUG_HIP_DYNAMIC_QUEUES=1" "AMD_OPT_FLUSH=0" "AMD_OPT_FLUSH=1" "GPU_FLUSH_ON_EXECUTION=1" "DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0" "DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1" "GPU_NUM_MEM_DEPENDENCY=0" "X=1"; do
    echo "$s ms/step $(run $s)"
  done
done | tee -a $OUT/runtime_knobs.txt
true
