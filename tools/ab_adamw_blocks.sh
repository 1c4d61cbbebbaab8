#!/bin/bash
# k_adamw duration in the replayed step against the number of workgroups of its launch (PVD_ADAMW_BLOCKS), rocprofv3 kernel traces.
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/${TAG:-r03blocks}; mkdir -p $OUT; export TMPDIR=/tmp
for b in ${BLOCKS:-1024 2048 4096 8192 16384}; do
  export PVD_ADAMW_BLOCKS=$b
  (cd /tmp && rm -rf /tmp/prof_k && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_k -o t -- python "$GRAFT_REPO_ROOT/bench.py" --steps 40 --warmup 20 --teacher-pretrain 0 --no-cpu-baseline > /tmp/prof_k.log 2>&1)
  T=$(find /tmp/prof_k -name "*kernel_trace.csv" | head -1)
  echo "== PVD_ADAMW_BLOCKS=$b: $(grep '^{' /tmp/prof_k.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.4f' % d['ms_per_step'])")" | tee -a $OUT/adamw_blocks.txt
  python tools/kernel_populations.py $T "k_adamw(" | tee -a $OUT/adamw_blocks.txt
done
