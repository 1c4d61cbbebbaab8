cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; OUT=gpurun_out/r03t; mkdir -p $OUT
PVD_PIPELINE_FORK=deep timeout 300 python -m pytest tests/test_hip_graph.py tests/test_hip_amp_parity.py -m gpu -q -p no:cacheprovider -k "not two_part" 2>&1 | tail -3
TAG=r03t tools/ab_env.sh PVD_PIPELINE_FORK start deep 2
export PVD_PIPELINE_FORK=deep
(cd /tmp && rm -rf /tmp/prof_d && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_d -o t -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > /tmp/prof_d.log 2>&1)
T=$(find /tmp/prof_d -name "*kernel_trace.csv" | head -1)
for k in k_hash_fwd_fused k_vm_bwd_split k_vm_fwd k_march_count_wave; do python tools/kernel_populations.py $T $k; done | tee $OUT/deep_populations.txt
python tools/step_timeline.py $T "k_adamw(" 22 2>&1 | tail -18 | cut -c1-120 | tee $OUT/deep_timeline.txt
