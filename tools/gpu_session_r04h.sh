#!/bin/bash
OUT=gpurun_out/r04h
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_fullsize.py tests/test_hip_bench_line.py -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest.txt
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; python - <<PY
import json
d=json.loads(open("$OUT/bench_line.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "sustained", d["sustained"], "frac", d["roofline"]["frac"])
print("psnr", json.dumps(d["psnr"], indent=1))
print("cpu", d["cpu_baseline"])
PY
timeout 300 python bench.py --workload teacher --steps 256 --warmup 320 > $OUT/bench_teacher.json 2>> $OUT/bench.err; cut -c1-300 $OUT/bench_teacher.json; python -c "
import json; d=json.loads(open('$OUT/bench_teacher.json').read().strip().splitlines()[-1]); print(d['cpu_baseline'])"
