#!/bin/bash
# round 6, GPU session 22: the PSNR comparison for the other students / an MLP teacher (no grid encoder on either side)
OUT=gpurun_out/r06s22
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python tools/psnr_vs_reference_kernels.py --student tensors > $OUT/psnr_hash_tensors.txt 2> $OUT/err1.txt; tail -3 $OUT/err1.txt; cat $OUT/psnr_hash_tensors.txt
timeout 1500 python tools/psnr_vs_reference_kernels.py --teacher-type mlp --student tensors > $OUT/psnr_mlp_tensors.txt 2> $OUT/err2.txt; tail -3 $OUT/err2.txt; cat $OUT/psnr_mlp_tensors.txt
timeout 1500 python tools/psnr_vs_reference_kernels.py --teacher-type mlp --student vm > $OUT/psnr_mlp_vm.txt 2> $OUT/err3.txt; tail -3 $OUT/err3.txt; cat $OUT/psnr_mlp_vm.txt
true
