#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { env "$@" timeout 300 python bench.py --steps 40 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['config']['launch'])"; }
run PVD_DP_FORCE=0
run PVD_DP_FORCE=1
run PVD_DP_FORCE=1 PVD_STEPS_PER_GRAPH=1
run PVD_DP_FORCE=1 PVD_STEPS_PER_GRAPH=5
run PVD_DP_FORCE=1 PVD_ADAMW_LAZY=0
run PVD_DP_FORCE=1 PVD_DP_INGRAPH=0
