#!/bin/bash
# round 6, GPU session 1: the two new tests, the round's baseline line on this box, the forced one-rank ray-DP step with its kernel
# timeline (where do the 44 us of recording tax sit?), the CPU baseline's thread scaling taken apart
OUT=gpurun_out/r06s1
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_bench_line.py::test_gpus_flag_alone_spawns_the_ranks tests/test_hip_fused_misc.py::test_flat_adamw_half_gradient_equals_widen_and_add -x -q 2>&1 | tail -5 | tee $OUT/tests.log
timeout 600 python bench.py --no-cpu-baseline --no-psnr > $OUT/bench_nocpu.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench_nocpu.json
PVD_DP_FORCE=1 PVD_DP_PIPELINE=2 timeout 600 python bench.py --no-cpu-baseline --no-psnr > $OUT/bench_dp1.json 2>> $OUT/bench.err; cut -c1-300 $OUT/bench_dp1.json
(cd /tmp && rm -rf /tmp/prof_dp && PVD_DP_FORCE=1 PVD_DP_PIPELINE=2 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dp -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --sustained-steps 0 > /tmp/prof_dp.log 2>&1)
T=$(find /tmp/prof_dp -name "*kernel_trace.csv" | head -1)
cp $(find /tmp/prof_dp -name "*kernel_stats.csv" | head -1) $OUT/dp1_kernel_stats.csv
python tools/step_timeline.py $T "k_adamw(" 30 > $OUT/dp1_step_timeline.txt 2>&1; tail -40 $OUT/dp1_step_timeline.txt
(cd /tmp && rm -rf /tmp/prof_b && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --sustained-steps 0 > /tmp/prof_b.log 2>&1)
T=$(find /tmp/prof_b -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $T "k_adamw(" 22 > $OUT/step_timeline.txt 2>&1; tail -26 $OUT/step_timeline.txt
timeout 900 python tools/cpu_baseline_scaling.py --counts 1,16,32,64,128 --steps 2 2>&1 | grep -v amdgpu | tee $OUT/cpu_baseline_scaling.txt
true
