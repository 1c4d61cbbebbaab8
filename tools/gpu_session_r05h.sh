#!/bin/bash
# round 5, session f: k_head_bwd with two tiles per trip (NT = 2: shared weight fragments, K = 32 weight-gradient MFMAs) against one
OUT=gpurun_out/r05h
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=$PWD/aaai2023-pvd_amd
echo "== parity"; timeout 900 python -m pytest tests/test_hip_head.py tests/test_hip_amp_parity.py tests/test_hip_amp_oracle.py tests/test_hip_golden.py -m gpu -q -s 2>&1 | grep -E "bit-identical|passed|failed|Error|error|assert" | tee $OUT/pytest_head.txt
echo "== stamps (nt2)"; PVD_HIP_LIB=$P/libpvd_hip_hprof.so timeout 200 python tools/prof_head_stamps.py 2>&1 | grep -v amdgpu.ids | tee $OUT/stamps_nt2.txt
echo "== stamps (nt1)"; PVD_HIP_LIB=$P/libpvd_hip_hprof_nt1.so timeout 200 python tools/prof_head_stamps.py 2>&1 | grep -v amdgpu.ids | tee $OUT/stamps_nt1.txt
echo "== alone (nt2)";  timeout 300 python tools/bench_head.py 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_head_nt2.txt
echo "== alone (nt1)"; PVD_HIP_LIB=$P/libpvd_hip_nt1.so timeout 300 python tools/bench_head.py 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_head_nt1.txt
echo "== the step"
for lib in "" $P/libpvd_hip_nt1.so "" $P/libpvd_hip_nt1.so "" $P/libpvd_hip_nt1.so; do PVD_HIP_LIB=$lib timeout 300 python bench.py --no-psnr --no-cpu-baseline 2>>$OUT/bench.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('${lib:-nt2 (in-tree)}'.split('/')[-1], 'ms/step %.4f' % d['ms_per_step'], 'loss %.5f' % d['config'].get('loss', -1))" | tee -a $OUT/step_ab.txt; done
echo "== teacher step"
for lib in "" $P/libpvd_hip_nt1.so; do PVD_HIP_LIB=$lib timeout 300 python bench.py --workload teacher --steps 256 --warmup 320 --no-cpu-baseline 2>>$OUT/bench.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('${lib:-nt2 (in-tree)}'.split('/')[-1], 'teacher ms/step %.4f' % d['ms_per_step'])" | tee -a $OUT/step_ab.txt; done
echo "== full suite"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
