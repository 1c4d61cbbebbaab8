#!/bin/bash
# round 5, session b: after the removal of the rejected variants -- full GPU suite, the bench line, the atomic probe's locality rows
OUT=gpurun_out/r05b
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 120 tools/probes/atomic_rate 2>&1 | grep -E "seq-rows|16KB-win|4KB-runs|16-lanes|^agent .* 8192 .*270000" | tee $OUT/atomic_rate_locality.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
for i in 1 2; do timeout 300 python bench.py --no-psnr --no-cpu-baseline 2>>$OUT/bench.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ms/step %.4f' % d['ms_per_step'], 'loss %.5f' % d['config'].get('loss', -1), 'frac', d['roofline']['frac'])" | tee -a $OUT/step.txt; done
timeout 300 python bench.py --steps 20 --warmup 5 --no-psnr --no-cpu-baseline 2>>$OUT/bench.err | tail -1 | cut -c1-400 | tee -a $OUT/step.txt
tail -5 $OUT/bench.err
