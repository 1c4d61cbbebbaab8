#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --teacher-pretrain 100 $ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$ARGS $*', d['ms_per_step'], d['config']['loss'], d['config']['psnr_student_vs_teacher_db'], d['config']['launch'][:60])"; }
ARGS="--student hash"
run PVD_PIPELINE_INGRAPH=0
run PVD_PIPELINE_INGRAPH=1
run PVD_PIPELINE_INGRAPH=1 PVD_PIPELINE_FORK=optimizer
run PVD_PIPELINE_INGRAPH=0 PVD_STEPS_PER_GRAPH=1
run PVD_PIPELINE_INGRAPH=0
run PVD_PIPELINE_INGRAPH=1
