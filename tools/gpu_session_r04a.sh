#!/bin/bash
# Round-4 session A: the fused hash kernel after the offsets / index rewrite -- parity first, then the SOL table, then the step.
TAG=${TAG:-r04a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_hip_head.py tests/test_hip_reference_constants.py tests/test_hip_parity.py tests/test_hip_golden.py -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest_head.txt
timeout 400 python tools/hash_sol.py 2>&1 | tee $OUT/hash_sol.txt
for v in 7 0 14; do
  PVD_FUSED_VARIANT=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_v$v.json 2>> $OUT/bench.err; cut -c1-330 $OUT/bench_v$v.json
done
PVD_FUSED_VARIANT=7 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_v7_200.json 2>> $OUT/bench.err; cut -c1-330 $OUT/bench_v7_200.json
tail -5 $OUT/bench.err
