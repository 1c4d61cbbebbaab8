#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02lazy
timeout 900 python -m pytest tests/test_hip_fused_misc.py tests/test_hip_graph.py tests/test_hip_amp_parity.py tests/test_hip_workloads.py tests/test_hip_dp_graph.py tests/test_hip_golden.py -q -x 2>&1 | tail -5
for f in 0 1 0 1; do
  PVD_ADAMW_LAZY=$f timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lazy=$f', d['ms_per_step'], d['config']['capture_fallback'], d['config']['loss'], d['config']['psnr_student_vs_teacher_db'])" | tee -a gpurun_out/r02lazy/lazy.txt
done
for f in 0 1; do
  PVD_ADAMW_LAZY=$f timeout 300 python bench.py --student tensors --no-cpu-baseline --teacher-pretrain 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tensors lazy=$f', d['ms_per_step'], d['config']['capture_fallback'], d['config']['loss'])" | tee -a gpurun_out/r02lazy/lazy.txt
done
