#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline $ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$ARGS $*', d['ms_per_step'], d['config']['loss'], d['config']['psnr_student_vs_teacher_db'], d['config']['launch'][:60])"; }
run PVD_PIPELINE_FORK=split
run PVD_PIPELINE_FORK=mid
run PVD_PIPELINE_FORK=split
run PVD_PIPELINE_FORK=mid
ARGS="--student tensors --teacher-pretrain 100"
run PVD_PIPELINE_FORK=split
run PVD_PIPELINE_FORK=mid
