#!/bin/bash
# round 5, session a: the decoupled VM kernels -- parity, A/B against the walks, the memory-side atomic rate, the step
OUT=gpurun_out/r05a
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
echo "== atomic rate probe"; timeout 120 tools/probes/atomic_rate 2>&1 | tee $OUT/atomic_rate.txt
echo "== parity (decoupled kernels are the default)"
timeout 600 python -m pytest tests/test_hip_vm.py -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest_vm.txt
PVD_VM_KERNEL=wide timeout 600 python -m pytest tests/test_hip_vm.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest_vm_wide.txt
echo "== A/B alone"
timeout 300 python tools/bench_vm_ab.py 2>&1 | tee $OUT/vm_ab.txt
for cb in 16 32; do PVD_VM_CHUNK_BWD=$cb CAMS=0,2 KERNELS=0,1 timeout 200 python tools/bench_vm_ab.py 2>&1 | tee -a $OUT/vm_ab_chunks.txt; done
for cf in 32 64; do PVD_VM_CHUNK_FWD=$cf CAMS=0,2 KERNELS=0,1 timeout 200 python tools/bench_vm_ab.py 2>&1 | tee -a $OUT/vm_ab_chunks.txt; done
for lib in libpvd_hip_vm_16_2_2.so libpvd_hip_vm_8_2_4.so; do PVD_HIP_LIB=$PWD/aaai2023-pvd_amd/$lib CAMS=0,2 KERNELS=0,1 timeout 200 python tools/bench_vm_ab.py 2>&1 | tee -a $OUT/vm_ab_libs.txt; done
echo "== the step"
for k in pipe walk pipe walk; do PVD_VM_KERNEL=$k timeout 300 python bench.py --no-psnr --no-cpu-baseline 2>>$OUT/bench.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$k', 'ms/step %.4f' % d['ms_per_step'], 'loss %.5f' % d['config'].get('loss', -1))" | tee -a $OUT/step_ab.txt; done
tail -5 $OUT/bench.err
