#!/bin/bash
# round 6, GPU session 24: run-to-run scatter of the mlp teacher-training comparison (is -0.76 dB a difference or noise?)
OUT=gpurun_out/r06s24
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
timeout 900 python tools/psnr_vs_reference_kernels.py --teacher-training --teacher 3000 --teacher-type mlp 2>/dev/null | grep "^A \|^B \|difference" | tee -a $OUT/mlp_scatter.txt
done
for i in 1 2; do
timeout 900 python tools/psnr_vs_reference_kernels.py --teacher-training --teacher 3000 2>/dev/null | grep "^A \|^B \|difference" | tee -a $OUT/hash_scatter.txt
done
true
