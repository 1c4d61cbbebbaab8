#!/bin/bash
# round 6, GPU session 13: the reference's kernels timed next to this repo's; the whole GPU suite on the current build
OUT=gpurun_out/r06s13
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 600 python tools/bench_vs_reference_kernels.py 2>&1 | grep -v amdgpu | tee $OUT/vs_reference_kernels.txt
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest_gpu.log
true
