#!/bin/bash
# Duration of one kernel inside the replayed step for several builds of the library (PVD_HIP_LIB), from rocprofv3 kernel traces.
#   tools/ab_kernel_us.sh <kernel substring> <alt1.so> [alt2.so ...]      ("" = the in-tree build, always measured last)
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/${TAG:-r03kern}; mkdir -p $OUT; export TMPDIR=/tmp
K=$1; shift
for lib in "$@" ""; do
  name=${lib:-in-tree}
  [ -n "$lib" ] && export PVD_HIP_LIB=$PWD/aaai2023-pvd_amd/$lib || unset PVD_HIP_LIB
  (cd /tmp && rm -rf /tmp/prof_k && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_k -o t -- python "$GRAFT_REPO_ROOT/bench.py" --steps 40 --warmup 20 --teacher-pretrain 50 --no-cpu-baseline > /tmp/prof_k.log 2>&1)
  T=$(find /tmp/prof_k -name "*kernel_trace.csv" | head -1)
  echo "== $name: $(grep '^{' /tmp/prof_k.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.4f' % d['ms_per_step'])")" | tee -a $OUT/kernel_us.txt
  python tools/kernel_populations.py $T "$K" | tee -a $OUT/kernel_us.txt
done
