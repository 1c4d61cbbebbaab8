#!/bin/bash
# (ROC_SYSTEM_SCOPE_SIGNAL=0 HANGS the process on this image: never without a timeout)
# HIP runtime knobs against the replayed step (bench.py, 100 steps, 20 per graph): ms/step per setting, two rounds.
#   bash tools/ab_runtime_knobs.sh [out]   (GPU box; writes gpurun_out/<out>/runtime_knobs.txt)
OUT=${1:-r04k}
mkdir -p "$GRAFT_REPO_ROOT/gpurun_out/$OUT"
run() {
  env "$@" timeout 60 python bench.py --no-psnr --no-cpu-baseline --sustained-steps 0 2>/dev/null | python -c "
import sys, json
l = [x for x in sys.stdin if x.startswith('{')]
print('%.4f' % json.loads(l[0])['ms_per_step'] if l else 'FAILED')"
}
for round in 1; do
  for s in "X=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=2" "GPU_STREAMOPS_CP_WAIT=1" "X=1"; do
    echo "$s ms/step $(run $s)"
  done
done | tee -a "$GRAFT_REPO_ROOT/gpurun_out/$OUT/runtime_knobs.txt"
