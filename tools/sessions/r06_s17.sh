#!/bin/bash
OUT=gpurun_out/r06s17
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_hip_reference_kernels.py -q 2>&1 | tail -12 | tee $OUT/tests.log
true
