#!/bin/bash
# round 6, GPU session 23: configs[1] (teacher training) through the reference's kernels + PyTorch vs libpvd_hip.so: held-out PSNR
OUT=gpurun_out/r06s23
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python tools/psnr_vs_reference_kernels.py --teacher-training --teacher 3000 > $OUT/psnr_teacher_hash.txt 2> $OUT/err1.txt; tail -3 $OUT/err1.txt; cat $OUT/psnr_teacher_hash.txt
timeout 900 python tools/psnr_vs_reference_kernels.py --teacher-training --teacher 3000 --teacher-type mlp > $OUT/psnr_teacher_mlp.txt 2> $OUT/err2.txt; tail -3 $OUT/err2.txt; cat $OUT/psnr_teacher_mlp.txt
true
