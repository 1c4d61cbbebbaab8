#!/bin/bash
# skew experiment + PMC of the SOL rows and the product kernel
OUT=gpurun_out/r04d
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for sk in 0 20 40 60 80 120; do
  echo "== skew $sk"; PVD_FUSED_SKEW=$sk python tools/hash_sol.py --rows "fused), default" 2>&1 | tail -1
done | tee $OUT/ab_skew.txt
for sk in 40 80; do
echo "== stamps skew $sk"; PVD_FUSED_SKEW=$sk PVD_HIP_LIB=$GRAFT_REPO_ROOT/aaai2023-pvd_amd/libpvd_hip_prof.so timeout 200 python tools/prof_fused_stamps.py 2>&1 | grep -v amdgpu.ids | tail -14
done | tee $OUT/stamps_skew.txt
# PMC: TCC request / hit / miss and the L1 side, separate passes
: > $OUT/pmc_sol.csv
for row in "product: lookup + head (fused), default" "gather G=14 (one round trip)" "gather G=14, all-hit tables" "gather levels 10-13 only" "stream 516 B/sample, 2048 wg"; do
  tag=$(echo "$row" | tr -c 'A-Za-z0-9' '_' | cut -c1-40)
  for c in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum"; do
    n=$(echo $c | cut -d' ' -f1)
    (cd /tmp && rm -rf /tmp/pmc_x && timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_x -- python "$GRAFT_REPO_ROOT/tools/hash_sol.py" --pmc "$row" > /tmp/pmc_x.log 2>&1)
    f=$(find /tmp/pmc_x -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python tools/pmc_summary.py $f | grep -E "k_hash_fwd_fused|k_sol_|^kernel" | sed "s/^/$tag,/" >> $OUT/pmc_sol.csv; else echo "$tag,$n,NO OUTPUT" >> $OUT/pmc_sol.csv; tail -3 /tmp/pmc_x.log >> $OUT/pmc_err.txt; fi
  done
done
cat $OUT/pmc_sol.csv
