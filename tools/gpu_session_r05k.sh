#!/bin/bash
# round 5, session k: ray-DP -- the forced one-rank RCCL step, the wire format's PSNR on two gloo ranks
OUT=gpurun_out/r05k
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for m in "" "PVD_DP_FORCE=1 PVD_DP_PIPELINE=2"; do for i in 1 2; do env $m timeout 300 python bench.py --no-psnr --no-cpu-baseline 2>>$OUT/bench.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('${m:-single GPU}', 'ms/step %.4f' % d['ms_per_step'])" | tee -a $OUT/dp_step.txt; done; done
timeout 1500 python tools/dp_wire_psnr.py 2>&1 | grep -v "amdgpu.ids" | tail -12 | tee $OUT/dp_wire_psnr.txt
