#!/bin/bash
TAG=${TAG:-r02defer}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_hip_fused_misc.py tests/test_hip_graph.py tests/test_hip_amp_parity.py tests/test_hip_dp_graph.py tests/test_hip_occupancy.py -q -x 2>&1 | tail -5
for f in 0 1 0 1; do
  PVD_LOSS_DEFER=$f timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('defer=$f', d['ms_per_step'], d['config']['launch'], d['config']['capture_fallback'], d['config']['loss'])" | tee -a $OUT/defer.txt
done
