#!/bin/bash
OUT=gpurun_out/r06s16
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_bench_line.py -q -k "single_gpu" 2>&1 | tail -4 | tee $OUT/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.txt 2> $OUT/bench.err
grep '^{' $OUT/bench_driver_args.txt | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print(d['ms_per_step'], d['cpu_baseline']['value'], d['gpu_reference'])"
true
