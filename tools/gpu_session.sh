#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_edges.py tests/test_hip_budget.py tests/test_hip_workloads.py tests/test_hip_render_parity.py -q -x 2>&1 | tail -4
run() { env "$@" timeout 300 python bench.py --student hash --no-cpu-baseline --teacher-pretrain 100 --bound 2 --scene-scale 1.9 --dt-gamma 0.00390625 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['config']['samples_per_step_per_gpu'], d['config']['loss'])"; }
run PVD_MARCH_THREAD_PER_RAY=0
run PVD_MARCH_THREAD_PER_RAY=1
