#!/bin/bash
# round 6, GPU session 21: the new PSNR test, five times (scatter of the two bars)
OUT=gpurun_out/r06s21
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4 5; do
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $OUT/psnr_test_scatter.txt
import sys, types
sys.path[:0] = [".", "aaai2023-pvd_amd", "tools", "tests"]
from psnr_vs_reference_kernels import compare
from pvd.trainer import psnr
runs, _ = compare(types.SimpleNamespace(teacher=300, stage1=60, stage2=150, steps=400, student="vm"), which=("A", "B"))
(_, ra, ia, ta, sa), (_, rb, ib, tb, sb) = runs
print("A", ra.mean(0), "B", rb.mean(0), "B vs A renders %.2f dB" % float(psnr(ib, ia)), "wall %.1f / %.1f s" % (ta, tb))
PY
done
timeout 900 python -m pytest tests/test_hip_reference_kernels.py -q 2>&1 | tail -3 | tee $OUT/tests.log
true
