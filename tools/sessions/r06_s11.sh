#!/bin/bash
# round 6, GPU session 11: golden vectors from the reference's own kernels (oracle/_ref built in the container) + the parity report
OUT=gpurun_out/r06s11
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
ls -la oracle/_ref/ | tee $OUT/ref_files.txt
PVD_GOLDEN_OUT=$OUT timeout 600 python tests/golden/make_golden_ref_kernels.py 2>&1 | grep -v amdgpu | tail -5 | tee $OUT/make.log
cp $OUT/reference_kernels.npz tests/golden/reference_kernels.npz
timeout 600 python tools/ref_kernels_report.py 2>&1 | grep -v amdgpu | tee $OUT/ref_kernels_report.txt
true
