#!/bin/bash
# round 6, GPU session 9: the exchange forms for a student without deferred rows (Plenoxel), its one-rank step, the MFMA-busy PMC pass
OUT=gpurun_out/r06s9
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
line() { grep '^{' "$1" | tail -1; }
timeout 1800 python -m pytest tests/test_hip_dp_exchange.py tests/test_hip_budget.py -q 2>&1 | tail -8 | tee $OUT/tests.log
for m in classic allreduce sharded; do
  PVD_DP_FORCE=1 PVD_DP_PIPELINE=2 PVD_DP_EXCHANGE=$m timeout 600 python bench.py --teacher mlp --student tensors --data-type llff --teacher-pretrain 0 --no-cpu-baseline --no-psnr > $OUT/b_$m.txt 2>> $OUT/bench.err
  line $OUT/b_$m.txt | python -c "import sys,json;d=json.loads(sys.stdin.read());print('mlp->tensors llff, one-rank RCCL, $m', round(d['ms_per_step'],4), round(d['sustained']['ms_per_step'],4), d['config'].get('exchange','')[:90])" | tee -a $OUT/dp1_tensors.txt
done
timeout 600 python bench.py --teacher mlp --student tensors --data-type llff --teacher-pretrain 0 --no-cpu-baseline --no-psnr > $OUT/b_single.txt 2>> $OUT/bench.err
line $OUT/b_single.txt | python -c "import sys,json;d=json.loads(sys.stdin.read());print('mlp->tensors llff, plain single GPU', round(d['ms_per_step'],4), round(d['sustained']['ms_per_step'],4))" | tee -a $OUT/dp1_tensors.txt
for m in classic allreduce; do
  PVD_DP_FORCE=1 PVD_DP_PIPELINE=2 PVD_DP_EXCHANGE=$m timeout 600 python bench.py --student hash --teacher-pretrain 100 --no-cpu-baseline --no-psnr > $OUT/h_$m.txt 2>> $OUT/bench.err
  line $OUT/h_$m.txt | python -c "import sys,json;d=json.loads(sys.stdin.read());print('hash->hash, one-rank RCCL, $m', round(d['ms_per_step'],4), round(d['sustained']['ms_per_step'],4), d['config'].get('exchange','')[:90])" | tee -a $OUT/dp1_tensors.txt
done
(cd /tmp && rm -rf /tmp/pmcs_mfma && timeout 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmcs_mfma -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 5 --teacher-pretrain 20 --no-cpu-baseline --no-psnr --sustained-steps 0 --eager > /tmp/pmcs_mfma.log 2>&1)
f=$(find /tmp/pmcs_mfma -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python tools/pmc_summary.py $f | grep -E "k_head_bwd|k_head_fwd|k_hash_fwd_fused|^kernel" | tee $OUT/pmc_mfma.csv
tail -3 /tmp/pmcs_mfma.log
true
