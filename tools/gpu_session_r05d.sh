#!/bin/bash
# round 5, session d: k_head_bwd's transposes as ds_write_b64 + ds_read_b64_tr_b16 (swizzled) against the four 2-byte stores + read
OUT=gpurun_out/r05d
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=$PWD/aaai2023-pvd_amd
timeout 300 python -m pytest tests/test_hip_graph.py -m gpu -x -q -k forked_recording 2>&1 | grep -v "^  warnings\|amdgpu.ids" | tail -70 | tee $OUT/pytest_fallback.txt
echo "== parity"; timeout 900 python -m pytest tests/test_hip_head.py tests/test_hip_amp_parity.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest_head.txt
echo "== stamps (tr)";  PVD_HIP_LIB=$P/libpvd_hip_hprof.so timeout 200 python tools/prof_head_stamps.py 2>&1 | grep -v amdgpu.ids | tee $OUT/stamps_tr.txt
echo "== stamps (b16)"; PVD_HIP_LIB=$P/libpvd_hip_hprof_tr0.so timeout 200 python tools/prof_head_stamps.py 2>&1 | grep -v amdgpu.ids | tee $OUT/stamps_b16.txt
echo "== alone (tr)";  timeout 300 python tools/bench_head.py 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_head_tr.txt
echo "== alone (b16)"; PVD_HIP_LIB=$P/libpvd_hip_tr0.so timeout 300 python tools/bench_head.py 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_head_b16.txt
echo "== the step"
for lib in "" $P/libpvd_hip_tr0.so "" $P/libpvd_hip_tr0.so "" $P/libpvd_hip_tr0.so; do PVD_HIP_LIB=$lib timeout 300 python bench.py --no-psnr --no-cpu-baseline 2>>$OUT/bench.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('${lib:-tr (in-tree)}'.split('/')[-1], 'ms/step %.4f' % d['ms_per_step'], 'loss %.5f' % d['config'].get('loss', -1))" | tee -a $OUT/step_ab.txt; done
