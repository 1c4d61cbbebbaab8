#!/bin/bash
TAG=${TAG:-r02i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log
timeout 300 python tools/bench_grid_bwd2.py > $OUT/bench_grid_bwd.log 2>&1; cat $OUT/bench_grid_bwd.log
timeout 300 python bench.py --workload teacher --steps 64 --warmup 16 > $OUT/bench_teacher.json 2> $OUT/bench_teacher.err; tail -2 $OUT/bench_teacher.err; cat $OUT/bench_teacher.json
