#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline $ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$ARGS $*', d['ms_per_step'], d['config']['loss'], d['config']['psnr_student_vs_teacher_db'], d['config']['launch'][:60])"; }
ARGS="--steps 20 --warmup 5"
run PVD_PIPELINE_CARRY=0
run PVD_PIPELINE_CARRY=1
run PVD_PIPELINE_CARRY=0
run PVD_PIPELINE_CARRY=1
ARGS=""
run PVD_PIPELINE_CARRY=0
run PVD_PIPELINE_CARRY=1
timeout 900 python -m pytest tests/test_hip_dp_graph.py tests/test_hip_graph.py tests/test_hip_workloads.py tests/test_hip_amp_parity.py -x -q -m gpu 2>&1 | tail -3
