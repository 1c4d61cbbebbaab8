#!/bin/bash
# round 6, GPU session 6: the whole GPU suite + the driver's command + smoke on the round's mid-point build
OUT=gpurun_out/r06s6
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.txt 2> $OUT/bench.err; grep '^{' $OUT/bench_driver_args.txt | tail -1 | cut -c1-400
true
