#!/bin/bash
# round 6, GPU session 15: the metric's step through the reference's own kernels + PyTorch, next to this repo's
OUT=gpurun_out/r06s15
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python tools/bench_reference_kernels_step.py 2>&1 | grep -v amdgpu | tail -12 | tee $OUT/reference_kernels_step.txt
timeout 900 python tools/bench_reference_kernels_step.py --student tensors 2>&1 | grep -v amdgpu | tail -5 | tee -a $OUT/reference_kernels_step.txt
true
