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
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench_line.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --sustained-steps 0 > /tmp/prof_b.log 2>&1)
T=$(find /tmp/prof_b -name "*kernel_trace.csv" | head -1)
cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
grep '^{' /tmp/prof_b.log | tail -1 > $OUT/bench_profiled_line.json
python tools/kernel_populations.py $T k_hash_fwd_fused > $OUT/kernel_populations.txt
for k in k_vm_bwd_split k_vm_fwd "k_adamw(" k_head_bwd; do python tools/kernel_populations.py $T "$k" >> $OUT/kernel_populations.txt; done
python tools/step_queues.py $T k_vm_bwd_split 22 > $OUT/step_timeline.txt 2>&1; cat $OUT/step_timeline.txt | cut -c1-100
python tools/in_step_record.py $OUT/kernel_populations.txt $OUT/bench_profiled_line.json > $OUT/in_step.json; cat $OUT/in_step.json
python tools/roofline_from_profile.py $OUT/kernel_populations.txt $OUT/bench_profiled_line.json $OUT/kernel_stats.csv | tee $OUT/roofline_rederived.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_driver_args.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
true
