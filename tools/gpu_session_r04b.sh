#!/bin/bash
TAG=${TAG:-r04b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_hip_head.py tests/test_hip_infer_rounds.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest_head.txt
timeout 400 python tools/hash_sol.py --rows product 2>&1 | tee $OUT/hash_sol.txt
for v in 14 7 0; do
  PVD_FUSED_VARIANT=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_v$v.json 2>> $OUT/bench.err; python - <<PY
import json
d=json.loads(open("$OUT/bench_v$v.json").read().strip().splitlines()[-1])
print("variant $v", d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("alone"), d["roofline"].get("in_step"))
PY
done
