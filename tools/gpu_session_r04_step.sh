#!/bin/bash
# The headline workload's evidence only (after a change of the step's schedule that leaves the kernels alone): bench line with
# cpu_baseline / psnr / sustained, the driver's command under rocprofv3 --kernel-trace --stats -> kernel stats, populations,
# step timeline, in-step record, re-derived roofline; the driver-argument line.  The PMC passes and the other workloads' lines
# of tools/gpu_session_r04.sh are unaffected and not repeated.
TAG=${TAG:-r04ev2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
: > $OUT/bench.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --sustained-steps 0 > /tmp/prof_b.log 2>&1)
T=$(find /tmp/prof_b -name "*kernel_trace.csv" | head -1)
cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
grep '^{' /tmp/prof_b.log | tail -1 > $OUT/bench_profiled_line.json
python tools/kernel_populations.py $T k_hash_fwd_fused > $OUT/kernel_populations.txt
for k in k_vm_bwd_split k_vm_fwd "k_adamw(" k_head_bwd; do python tools/kernel_populations.py $T "$k" >> $OUT/kernel_populations.txt; done
python tools/step_queues.py $T k_vm_bwd_split 22 > $OUT/step_timeline.txt 2>&1; cat $OUT/step_timeline.txt | cut -c1-100
python tools/in_step_record.py $OUT/kernel_populations.txt $OUT/bench_profiled_line.json > $OUT/in_step.json; cat $OUT/in_step.json
python tools/roofline_from_profile.py $OUT/kernel_populations.txt $OUT/bench_profiled_line.json $OUT/kernel_stats.csv | tee $OUT/roofline_rederived.txt
if [ "${PMC:-0}" = 1 ]; then  # the roofline kernel's PMC passes (after a change of ITS sources: bench.py keys the record by their hash)
  for c in FETCH_SIZE WRITE_SIZE "TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum"; do
    n=$(echo $c | cut -d' ' -f1)
    (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$n -- python "$GRAFT_REPO_ROOT/tools/pmc_teacher_fwd.py" > /tmp/pmc_$n.log 2>&1)
  done
  n=$(grep samples_per_launch /tmp/pmc_FETCH_SIZE.log | awk '{print $2}')
  python tools/pmc_traffic_json.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $n > $OUT/pmc_traffic.json; cat $OUT/pmc_traffic.json
  f=$(find /tmp/pmc_TCC_REQ_sum -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f | grep -i "k_hash_fwd_fused\|^kernel" > $OUT/pmc_tcc.csv; cat $OUT/pmc_tcc.csv
  # the records bench.py quotes are keyed by the kernel sources' hash: install this session's before the lines below are produced
  cp $OUT/pmc_traffic.json profiles/r04_pmc_traffic.json; cp $OUT/in_step.json profiles/r04_in_step.json
fi
timeout 600 python bench.py > $OUT/bench_line.json 2>> $OUT/bench.err; cut -c1-300 $OUT/bench_line.json
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_driver_args.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
true
