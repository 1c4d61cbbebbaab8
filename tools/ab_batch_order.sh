#!/bin/bash
# Does an image-space Z order of the step's 4096 pixel ids (PVD_BATCH_ORDER=morton: the same rays, only the row order moves)
# make the table gathers / scatters more coherent?  Kernel durations from a rocprofv3 kernel trace of the replayed step, and the
# L2 request / miss counters of the lookup kernels from a PMC pass over eager steps, for both orders.
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/${TAG:-r03order}; mkdir -p $OUT; export TMPDIR=/tmp
for o in random morton; do
  [ $o = morton ] && export PVD_BATCH_ORDER=morton || unset PVD_BATCH_ORDER
  (cd /tmp && rm -rf /tmp/prof_$o && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$o -o t -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > /tmp/prof_$o.log 2>&1)
  T=$(find /tmp/prof_$o -name "*kernel_trace.csv" | head -1)
  echo "== pixel order: $o   ($(grep '^{' /tmp/prof_$o.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.4f, %d samples/step' % (d['ms_per_step'], d['config']['samples_per_step_per_gpu']))"))" | tee -a $OUT/batch_order.txt
  for k in k_hash_fwd_fused k_vm_fwd k_vm_bwd_split k_march_count_wave; do python tools/kernel_populations.py $T $k | tee -a $OUT/batch_order.txt; done
  (cd /tmp && rm -rf /tmp/pmc_$o && timeout 300 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_MISS_sum --output-format csv -d /tmp/pmc_$o -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 5 --teacher-pretrain 20 --no-cpu-baseline --eager > /tmp/pmc_$o.log 2>&1)
  f=$(find /tmp/pmc_$o -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f | grep -E "k_hash_fwd_fused|k_vm_fwd|k_vm_bwd_split" | tee -a $OUT/batch_order.txt
done
