cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r03d; mkdir -p $OUT
for lib in libpvd_hip_vmfwd_unbatched.so libpvd_hip.so; do echo "== $lib"; PVD_HIP_LIB=$PWD/aaai2023-pvd_amd/$lib timeout 200 python tools/bench_vm.py 2>&1 | grep -v amdgpu | head -12; done | tee $OUT/vm_fwd_ab.txt
timeout 900 python -m pytest tests/test_hip_fused_misc.py tests/test_hip_graph.py tests/test_hip_amp_parity.py tests/test_hip_vm.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -25 | tee $OUT/pytest.log
for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20"; do for sp in 0 1 0 1; do
  PVD_ADAMW_SPLIT=$sp timeout 300 python bench.py $args --no-cpu-baseline 2>>$OUT/err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('split=$sp', '$args', 'ms/step %.4f' % d['ms_per_step'], d['config']['launch'][16:120], 'loss %.4f psnr %.2f' % (d['config']['loss'], d['config']['psnr_student_vs_teacher_db']))" | tee -a $OUT/adamw_split_ab.txt
done; done
