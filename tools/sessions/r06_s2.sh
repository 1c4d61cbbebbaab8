#!/bin/bash
# round 6, GPU session 2: the ray-DP rework (objective riding under DP, gather-zero-check feeding the update, sharded update)
OUT=gpurun_out/r06s2
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_hip_dp_exchange.py -x -q 2>&1 | tail -25 | tee $OUT/t_exchange.log
timeout 1500 python -m pytest tests/test_hip_dp_graph.py -q 2>&1 | tail -25 | tee $OUT/t_dp_graph.log
timeout 900 python -m pytest tests/test_hip_fused_misc.py tests/test_hip_graph.py -q 2>&1 | tail -8 | tee $OUT/t_misc.log
for m in classic allreduce sharded; do
  # (PVD_DP_EXCHANGE=classic implies the separate objective launches since the knob PVD_DP_RIDE was folded into it)
  PVD_DP_FORCE=1 PVD_DP_PIPELINE=2 PVD_DP_EXCHANGE=$m timeout 600 python bench.py --no-cpu-baseline --no-psnr > $OUT/bench_dp1_$m.json 2>> $OUT/bench.err
  python -c "import json;d=json.load(open('$OUT/bench_dp1_$m.json'));print('$m', d['ms_per_step'], d['sustained']['ms_per_step'], d['config'].get('exchange'))" | tee -a $OUT/dp1_modes.txt
done
timeout 600 python bench.py --no-cpu-baseline --no-psnr > $OUT/bench_single.json 2>> $OUT/bench.err
python -c "import json;d=json.load(open('$OUT/bench_single.json'));print('single', d['ms_per_step'], d['sustained']['ms_per_step'])" | tee -a $OUT/dp1_modes.txt
(cd /tmp && rm -rf /tmp/prof_dp && PVD_DP_FORCE=1 PVD_DP_PIPELINE=2 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dp -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --sustained-steps 0 > /tmp/prof_dp.log 2>&1)
T=$(find /tmp/prof_dp -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $T "k_vm_bwd_split" 12 > $OUT/dp1_step_timeline.txt 2>&1; cat $OUT/dp1_step_timeline.txt
tail -5 $OUT/bench.err
true
