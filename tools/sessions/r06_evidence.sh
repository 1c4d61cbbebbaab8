#!/bin/bash
# The round's evidence run (lean: the GPU budget is 90 minutes a round).  TAG names the output directory under gpurun_out/.
#   bench line (default arguments, with cpu_baseline + psnr) -> rocprofv3 --kernel-trace --stats of the DRIVER's command
#   (bench.py --steps 20 --warmup 5) -> kernel stats, launch populations, step timeline -> PMC passes of the roofline kernel
#   (FETCH_SIZE, WRITE_SIZE; TCC request / miss counters) -> re-derivation -> the other workloads' lines -> PMC of the step's kernels.
TAG=${TAG:-r06ev}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
# PMC of the roofline kernel first: this run's bench line then carries `traffic` of THIS build
for c in FETCH_SIZE WRITE_SIZE "TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum TCC_EA0_RDREQ_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  (cd /tmp && rm -rf /tmp/pmc_$n && timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$n -- python "$GRAFT_REPO_ROOT/tools/pmc_teacher_fwd.py" > /tmp/pmc_$n.log 2>&1)
done
n=$(grep samples_per_launch /tmp/pmc_FETCH_SIZE.log | awk '{print $2}')
python tools/pmc_traffic_json.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $n > $OUT/pmc_traffic.json; cat $OUT/pmc_traffic.json
cp $OUT/pmc_traffic.json profiles/r06_pmc_traffic.json
f=$(find /tmp/pmc_TCC_REQ_sum -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python tools/pmc_summary.py $f | grep -i "k_hash_fwd_fused\|^kernel" > $OUT/pmc_tcc.csv; cat $OUT/pmc_tcc.csv
# the profiled run of the driver's command
(cd /tmp && rm -rf /tmp/prof_b && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --sustained-steps 0 > /tmp/prof_b.log 2>&1)
T=$(find /tmp/prof_b -name "*kernel_trace.csv" | head -1)
cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
grep '^{' /tmp/prof_b.log | tail -1 > $OUT/bench_profiled_line.json
python tools/kernel_populations.py $T k_hash_fwd_fused > $OUT/kernel_populations.txt
for k in k_vm_bwd_split k_vm_fwd "k_adamw(" k_head_bwd k_head_fwd k_composite_bwd_wave k_march_count_wave; do python tools/kernel_populations.py $T "$k" >> $OUT/kernel_populations.txt; done
cat $OUT/kernel_populations.txt
python tools/step_timeline.py $T "k_adamw(" 22 > $OUT/step_timeline.txt 2>&1; tail -24 $OUT/step_timeline.txt
python tools/in_step_record.py $OUT/kernel_populations.txt $OUT/bench_profiled_line.json > $OUT/in_step.json; cat $OUT/in_step.json
cp $OUT/in_step.json profiles/r06_in_step.json
python tools/roofline_from_profile.py $OUT/kernel_populations.txt $OUT/bench_profiled_line.json $OUT/kernel_stats.csv | tee $OUT/roofline_rederived.txt
# the line itself (default arguments; cpu_baseline on SURVEY 8(d)'s protocol, psnr)
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench_line.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_driver_args.json
timeout 300 python bench.py --workload teacher --steps 256 --warmup 320 > $OUT/bench_teacher.json 2>> $OUT/bench.err; cut -c1-250 $OUT/bench_teacher.json
timeout 300 python bench.py --student hash --no-cpu-baseline --no-psnr --teacher-pretrain 100 > $OUT/bench_hash_student.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_hash_student.json
timeout 300 python bench.py --student tensors --no-cpu-baseline --no-psnr --teacher-pretrain 100 > $OUT/bench_tensors_student.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_tensors_student.json
timeout 300 python bench.py --teacher mlp --student tensors --data-type llff --no-cpu-baseline --no-psnr --teacher-pretrain 0 > $OUT/bench_config3_mlp_tensors_llff.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_config3_mlp_tensors_llff.json
timeout 300 python bench.py --student hash --data-type tank --bound 2 --dt-gamma 0.00390625 --scene-scale 1.9 --no-cpu-baseline --no-psnr --teacher-pretrain 100 > $OUT/bench_config4_hash_hash_tank.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_config4_hash_hash_tank.json
timeout 300 python tools/bench_render.py 2>&1 | grep -v amdgpu > $OUT/render.txt; cat $OUT/render.txt
# the lookup's split-map probe (tools/probes/hash_sol.hip: k_sol_split) against the plain gather of the same four levels: fabric reads
: > $OUT/pmc_hash_sol_split.csv
i=0
for row in "gather levels 10-13 only" "split map: levels 10-13, pair-owned rows, 1456 wg"; do
  i=$((i + 1))
  (cd /tmp && rm -rf /tmp/pmc_sol$i && timeout 200 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum TCC_EA0_RDREQ_sum --output-format csv -d /tmp/pmc_sol$i -- python "$GRAFT_REPO_ROOT/tools/hash_sol.py" --pmc "$row" > /tmp/pmc_sol$i.log 2>&1)
  f=$(find /tmp/pmc_sol$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f | grep -i "k_sol\|^kernel" | sed "s/^/$row,/" >> $OUT/pmc_hash_sol_split.csv
done
cat $OUT/pmc_hash_sol_split.csv
# where the step's kernels spend their wave cycles (separate PMC passes over eager steps)
bash tools/pmc_step_kernels.sh $TAG
true
