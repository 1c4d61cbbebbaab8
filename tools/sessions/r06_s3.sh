#!/bin/bash
# round 6, GPU session 3: the ride under ray-DP as an all-reduce of fixed-size partials; one-rank step by exchange form
OUT=gpurun_out/r06s3
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
line() { grep '^{' "$1" | tail -1; }
timeout 900 python -m pytest tests/test_hip_fused_misc.py -q -k "objective" 2>&1 | tail -4 | tee $OUT/t_ride.log
timeout 1500 python -m pytest tests/test_hip_dp_graph.py -q 2>&1 | tail -6 | tee $OUT/t_dp_graph.log
for m in classic allreduce sharded; do
  # (PVD_DP_EXCHANGE=classic implies the separate objective launches since the knob PVD_DP_RIDE was folded into it)
  PVD_DP_FORCE=1 PVD_DP_PIPELINE=2 PVD_DP_EXCHANGE=$m timeout 600 python bench.py --no-cpu-baseline --no-psnr > $OUT/bench_dp1_$m.txt 2>> $OUT/bench.err
  line $OUT/bench_dp1_$m.txt > $OUT/bench_dp1_$m.json
  python -c "import json;d=json.load(open('$OUT/bench_dp1_$m.json'));print('$m', d['ms_per_step'], d['sustained']['ms_per_step'])" | tee -a $OUT/dp1_modes.txt
done
timeout 600 python bench.py --no-cpu-baseline --no-psnr > $OUT/bench_single.json 2>> $OUT/bench.err
python -c "import json;d=json.load(open('$OUT/bench_single.json'));print('single', d['ms_per_step'], d['sustained']['ms_per_step'])" | tee -a $OUT/dp1_modes.txt
(cd /tmp && rm -rf /tmp/prof_dp && PVD_DP_FORCE=1 PVD_DP_PIPELINE=2 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dp -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --sustained-steps 0 > /tmp/prof_dp.log 2>&1)
T=$(find /tmp/prof_dp -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $T "k_vm_bwd_split" 12 > $OUT/dp1_step_timeline.txt 2>&1; cat $OUT/dp1_step_timeline.txt
true
