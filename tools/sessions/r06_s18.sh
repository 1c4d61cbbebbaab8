#!/bin/bash
OUT=gpurun_out/r06s18
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_hip_reference_kernels.py -q -k "random_occupancy" 2>&1 | grep -E "^E  |FAILED|passed|failed|cfg" | head -60 | tee $OUT/tests.log
true
