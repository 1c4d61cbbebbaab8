#!/bin/bash
# round 5, session l: the lookup's split-map probe row; separate vs interleaved VM tables
OUT=gpurun_out/r05l
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/hash_sol.py 2>&1 | grep -v amdgpu.ids | tee $OUT/hash_sol_table.txt
timeout 300 python tools/bench_vm.py 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_vm.txt
