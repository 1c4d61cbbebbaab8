#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { env "$@" timeout 300 python bench.py --workload teacher --steps 256 --warmup 320 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['config'].get('loss'), d['config'].get('launch'))"; }
run PVD_TEACHER_PIPELINE=0
run PVD_TEACHER_PIPELINE=1
run PVD_TEACHER_PIPELINE=0
run PVD_TEACHER_PIPELINE=1
timeout 900 python -m pytest tests/test_hip_budget.py -x -q -m gpu 2>&1 | tail -3
