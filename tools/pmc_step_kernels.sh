#!/bin/bash
# Where the step's big kernels spend their wave cycles (separate rocprofv3 --pmc passes over eager steps of bench.py):
#   bash tools/pmc_step_kernels.sh [out]   (GPU box; writes gpurun_out/<out>/pmc_step_kernels.csv)
OUT=${1:-r04n}
mkdir -p "$GRAFT_REPO_ROOT/gpurun_out/$OUT"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
: > gpurun_out/$OUT/pmc_step_kernels.csv
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "TCC_EA0_WRREQ_sum TCC_ATOMIC_sum TCC_REQ_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  (cd /tmp && rm -rf /tmp/pmcs_$n && timeout 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcs_$n -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 5 --teacher-pretrain 20 --no-cpu-baseline --no-psnr --sustained-steps 0 --eager > /tmp/pmcs_$n.log 2>&1)
  f=$(find /tmp/pmcs_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f | grep -E "k_head_bwd|k_head_fwd|k_vm_bwd_split|k_vm_fwd|k_adamw|k_composite|k_march|k_hash_fwd_fused|^kernel" >> gpurun_out/$OUT/pmc_step_kernels.csv
done
wc -l gpurun_out/$OUT/pmc_step_kernels.csv
