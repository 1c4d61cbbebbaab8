#!/bin/bash
OUT=gpurun_out/r05c
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tools/probes/tr16_probe | tee $OUT/tr16_probe.txt | head -20
for i in 1 2 3; do timeout 300 python -m pytest tests/test_hip_budget.py -m gpu -q -k teacher_block_graph 2>&1 | grep -E "passed|failed|allclose\(\[" | tee -a $OUT/budget_flake.txt; done
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_hip_budget.py::test_teacher_block_graph_follows_the_eager_run_across_grid_updates 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
