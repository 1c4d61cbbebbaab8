#!/bin/bash
# the evidence run (tests, smoke, bench line, rocprofv3 stats + launch populations + hardware queues + PMC, the other workloads);
# TAG names the output directory under gpurun_out/, PMC=0 / RENDER=0 skip the counter passes / the render and fused-MLP lines
TAG=${TAG:-r02final}
PMC=${PMC:-1}   # 0: skip the counter passes (profiles/r02_pmc_traffic.json stays valid while the lookup kernels' sources are unchanged)
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
if [ "$PMC" = 1 ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python "$GRAFT_REPO_ROOT/tools/pmc_teacher_fwd.py" > /tmp/pmc_$c.log 2>&1)
done
n=$(grep samples_per_launch /tmp/pmc_FETCH_SIZE.log | awk '{print $2}')
python tools/pmc_traffic_json.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $n > $OUT/pmc_traffic.json; cat $OUT/pmc_traffic.json
mkdir -p profiles; cp $OUT/pmc_traffic.json profiles/r02_pmc_traffic.json   # (so that this run's bench line can carry it: same build)
fi
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench_line.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline > /tmp/prof_b.log 2>&1)
cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
python tools/step_timeline.py $(find /tmp/prof_b -name "*kernel_trace.csv" | head -1) "k_adamw(" 22 > $OUT/step_timeline.txt; tail -3 $OUT/step_timeline.txt
T=$(find /tmp/prof_b -name "*kernel_trace.csv" | head -1)
python tools/kernel_populations.py $T k_hash_fwd_fused > $OUT/kernel_populations.txt; python tools/kernel_populations.py $T k_vm_bwd_split >> $OUT/kernel_populations.txt; cat $OUT/kernel_populations.txt
python tools/step_queues.py $T "k_adamw(" 22 > $OUT/step_queues.txt; tail -1 $OUT/step_queues.txt
grep '^{' /tmp/prof_b.log | tail -1 > $OUT/bench_profiled_line.json
[ "$PMC" = 1 ] && for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcs_$n -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 5 --teacher-pretrain 20 --no-cpu-baseline --eager > /tmp/pmcs_$n.log 2>&1)
  f=$(find /tmp/pmcs_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f | grep -i "pvd\|^kernel" >> $OUT/pmc_step_kernels.csv
done
[ "$PMC" = 1 ] && wc -l $OUT/pmc_step_kernels.csv
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_args.json 2>> $OUT/bench.err; cut -c1-250 $OUT/bench_driver_args.json
for m in "PVD_PIPELINE_INGRAPH=0" "PVD_DP_FORCE=1 PVD_DP_PIPELINE=2" "PVD_DP_FORCE=1 PVD_DP_PIPELINE=0"; do
  echo "$m: $(env $m timeout 300 python bench.py --no-cpu-baseline 2>> $OUT/bench.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['launch'][:70], d['config'].get('exchange','')[:60] if isinstance(d['config'].get('exchange'), str) else '')")" >> $OUT/schedules_ab.txt
done
cat $OUT/schedules_ab.txt
timeout 300 python bench.py --workload teacher --steps 256 --warmup 320 > $OUT/bench_teacher.json 2>> $OUT/bench.err; cut -c1-250 $OUT/bench_teacher.json
timeout 300 python bench.py --student hash --no-cpu-baseline --teacher-pretrain 100 > $OUT/bench_hash_student.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_hash_student.json
timeout 300 python bench.py --student tensors --no-cpu-baseline --teacher-pretrain 100 > $OUT/bench_tensors_student.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_tensors_student.json
timeout 300 python tools/bench_mlp_to_tensors.py 2>&1 | grep -v amdgpu > $OUT/bench_mlp_to_tensors.txt; cat $OUT/bench_mlp_to_tensors.txt
[ "${RENDER:-1}" = 1 ] && { timeout 300 python tools/bench_render.py 2>&1 | grep -v amdgpu > $OUT/bench_render.log; cat $OUT/bench_render.log; }
timeout 300 python tools/train_distill.py 2>&1 | grep -v amdgpu > $OUT/train_distill.txt; tail -3 $OUT/train_distill.txt
timeout 300 python bench.py --student hash --no-cpu-baseline --teacher-pretrain 100 --bound 2 --scene-scale 1.9 --dt-gamma 0.00390625 > $OUT/bench_hash_bound2.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_hash_bound2.json
[ "${RENDER:-1}" = 1 ] && { timeout 120 python tools/bench_mlp_fused.py 2>&1 | grep fused > $OUT/bench_mlp_fused.txt; cat $OUT/bench_mlp_fused.txt; }
true
