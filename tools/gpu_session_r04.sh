#!/bin/bash
# Round-4 evidence run (lean: the GPU budget is 90 minutes a round).  TAG names the output directory under gpurun_out/.
#   bench line (default arguments, with cpu_baseline) -> rocprofv3 --kernel-trace --stats of the DRIVER's command
#   (bench.py --steps 20 --warmup 5) -> kernel stats, launch populations of the roofline kernel and of the scatter, step
#   timeline -> PMC passes of the roofline kernel (FETCH_SIZE, WRITE_SIZE; TCC request / miss counters) -> re-derivation.
TAG=${TAG:-r04ev}
PMC=${PMC:-1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench_line.json
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --sustained-steps 0 > /tmp/prof_b.log 2>&1)
T=$(find /tmp/prof_b -name "*kernel_trace.csv" | head -1)
cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
grep '^{' /tmp/prof_b.log | tail -1 > $OUT/bench_profiled_line.json
python tools/kernel_populations.py $T k_hash_fwd_fused > $OUT/kernel_populations.txt
for k in k_vm_bwd_split k_vm_fwd "k_adamw(" k_head_bwd; do python tools/kernel_populations.py $T "$k" >> $OUT/kernel_populations.txt; done
cat $OUT/kernel_populations.txt
python tools/step_timeline.py $T "k_adamw(" 22 > $OUT/step_timeline.txt 2>&1; tail -30 $OUT/step_timeline.txt
python tools/in_step_record.py $OUT/kernel_populations.txt $OUT/bench_profiled_line.json > $OUT/in_step.json; cat $OUT/in_step.json
python tools/roofline_from_profile.py $OUT/kernel_populations.txt $OUT/bench_profiled_line.json $OUT/kernel_stats.csv | tee $OUT/roofline_rederived.txt
if [ "$PMC" = 1 ]; then
  for c in FETCH_SIZE WRITE_SIZE "TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum"; do
    n=$(echo $c | cut -d' ' -f1)
    (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$n -- python "$GRAFT_REPO_ROOT/tools/pmc_teacher_fwd.py" > /tmp/pmc_$n.log 2>&1)
  done
  n=$(grep samples_per_launch /tmp/pmc_FETCH_SIZE.log | awk '{print $2}')
  python tools/pmc_traffic_json.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $n > $OUT/pmc_traffic.json; cat $OUT/pmc_traffic.json
  f=$(find /tmp/pmc_TCC_REQ_sum -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f | grep -i "k_hash_fwd_fused\|^kernel" > $OUT/pmc_tcc.csv; cat $OUT/pmc_tcc.csv
fi
true
# smoke + the other workloads' lines (documentation: BASELINE.md rows 2, 4, 5)
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_driver_args.json
timeout 300 python bench.py --workload teacher --steps 256 --warmup 320 > $OUT/bench_teacher.json 2>> $OUT/bench.err; cut -c1-250 $OUT/bench_teacher.json
timeout 300 python bench.py --student hash --no-cpu-baseline --no-psnr --teacher-pretrain 100 > $OUT/bench_hash_student.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_hash_student.json
timeout 300 python bench.py --student tensors --no-cpu-baseline --no-psnr --teacher-pretrain 100 > $OUT/bench_tensors_student.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_tensors_student.json
timeout 300 python bench.py --teacher mlp --student tensors --data-type llff --no-cpu-baseline --no-psnr --teacher-pretrain 0 > $OUT/bench_config3_mlp_tensors_llff.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_config3_mlp_tensors_llff.json
timeout 300 python bench.py --student hash --data-type tank --bound 2 --dt-gamma 0.00390625 --scene-scale 1.9 --no-cpu-baseline --no-psnr --teacher-pretrain 100 > $OUT/bench_config4_hash_hash_tank.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_config4_hash_hash_tank.json
timeout 300 python tools/bench_render.py 2>&1 | grep -v amdgpu > $OUT/render.txt; cat $OUT/render.txt
timeout 200 python tools/make_blender_scene.py /tmp/chair_blender --views 40 --res 200 2>&1 | tail -1
timeout 300 python bench.py --workload teacher --steps 256 --warmup 320 --no-cpu-baseline --data-root /tmp/chair_blender > $OUT/bench_teacher_provider.json 2>> $OUT/bench.err; cut -c1-420 $OUT/bench_teacher_provider.json
true
# where k_head_bwd's time goes: wave-parked vs issue-stalled vs active quad-cycles, MFMA busy, LDS conflicts --
# separate PMC passes over eager steps
if [ "${PMC_HEAD:-0}" = 1 ]; then
  : > $OUT/pmc_head_bwd.csv
  for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS"; do
    n=$(echo $c | cut -d' ' -f1)
    (cd /tmp && rm -rf /tmp/pmch_$n && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmch_$n -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 5 --teacher-pretrain 20 --no-cpu-baseline --no-psnr --sustained-steps 0 --eager > /tmp/pmch_$n.log 2>&1)
    f=$(find /tmp/pmch_$n -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py $f | grep -E "k_head_bwd|k_head_fwd|k_vm_bwd_split|k_vm_fwd|k_adamw|^kernel" >> $OUT/pmc_head_bwd.csv
  done
  cat $OUT/pmc_head_bwd.csv
fi
true
