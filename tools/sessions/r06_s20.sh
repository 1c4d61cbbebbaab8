#!/bin/bash
# round 6, GPU session 20: PSNR after a whole staged distillation run -- the reference's kernels + PyTorch against libpvd_hip.so
OUT=gpurun_out/r06s20
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 600 python tools/psnr_vs_reference_kernels.py --teacher 200 --stage1 50 --stage2 100 --steps 300 > $OUT/psnr_small.txt 2> $OUT/psnr_small.err; tail -5 $OUT/psnr_small.err; cat $OUT/psnr_small.txt
timeout 1500 python tools/psnr_vs_reference_kernels.py > $OUT/psnr.txt 2> $OUT/psnr.err; tail -5 $OUT/psnr.err; cat $OUT/psnr.txt
true
