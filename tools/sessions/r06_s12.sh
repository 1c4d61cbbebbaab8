#!/bin/bash
# round 6, GPU session 12: the trimmed fixture from the reference's kernels, the report on it, the new tests
OUT=gpurun_out/r06s12
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
PVD_GOLDEN_OUT=$OUT timeout 600 python tests/golden/make_golden_ref_kernels.py 2>&1 | grep -v amdgpu | tail -3 | tee $OUT/make.log
cp $OUT/reference_kernels.npz tests/golden/reference_kernels.npz
timeout 600 python tools/ref_kernels_report.py 2>&1 | grep -v amdgpu > $OUT/ref_kernels_report.txt; grep -c "100.0000%" $OUT/ref_kernels_report.txt; grep -v "100.0000%" $OUT/ref_kernels_report.txt | head -5
timeout 900 python -m pytest tests/test_hip_reference_kernels.py tests/test_oracle_ref_kernels.py -q 2>&1 | tail -6 | tee $OUT/tests.log
true
