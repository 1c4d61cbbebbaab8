#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['config']['loss'], d['config']['psnr_student_vs_teacher_db'])"; }
run PVD_STEPS_PER_GRAPH=10
run A=1
run PVD_STEPS_PER_GRAPH=10
run A=1
timeout 900 python -m pytest tests/test_hip_dp_graph.py tests/test_hip_graph.py -x -q -m gpu 2>&1 | tail -3
