#!/bin/bash
# round 6, GPU session 25: the mlp teacher-training comparison over seeds (initial weights + batches): is the product's deficit systematic?
OUT=gpurun_out/r06s25
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for s in 1 2 3 4 5; do
timeout 900 python tools/psnr_vs_reference_kernels.py --teacher-training --teacher 3000 --teacher-type mlp --seed $s 2>/dev/null | grep "^A \|^B \|difference" | sed "s/^/seed $s  /" | tee -a $OUT/mlp_seeds.txt
done
true
