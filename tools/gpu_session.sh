#!/bin/bash
# One gpurun call of round 2 (edit per call).  Everything lands under gpurun_out/$TAG.
TAG=${TAG:-r02b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/bench_grid.py > $OUT/bench_grid.log 2>&1
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log; cat $OUT/bench_grid.log
