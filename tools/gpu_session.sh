#!/bin/bash
TAG=${TAG:-r02vmi}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/bench_vm.py 2>&1 | grep -v amdgpu > $OUT/bench_vm.txt; cat $OUT/bench_vm.txt
timeout 300 python -m pytest tests/test_hip_vm.py -q -x 2>&1 | tail -3
