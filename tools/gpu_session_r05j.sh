#!/bin/bash
# round 5, session j: the persistent VM render (pvd_infer_image_vm): parity with the round loops, time per 800 x 800 view
OUT=gpurun_out/r05j
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=$PWD/aaai2023-pvd_amd
timeout 900 python -m pytest tests/test_hip_infer_rounds.py -m gpu -q -x 2>&1 | grep -v "amdgpu.ids" | tail -5 | tee $OUT/pytest_infer.txt
for lib in "" $P/libpvd_hip_d2.so $P/libpvd_hip_d3.so $P/libpvd_hip_d6.so; do echo "== lib ${lib:-in-tree (depth 4)}"; PVD_HIP_LIB=$lib PVD_RENDER_ONLY=p PVD_RENDER_KIND=vm timeout 300 python tools/bench_render.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/render_depth.txt; done
