#!/bin/bash
# kernel-level breakdown of the one-launch renders (hash, VM, Plenoxel) (rocprofv3 --kernel-trace --stats over tools/bench_render.py, persistent mode only)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05q
(cd /tmp && rm -rf /tmp/prof_r && PVD_RENDER_ONLY=p timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r -o r -- python "$GRAFT_REPO_ROOT/tools/bench_render.py" > /tmp/prof_r.log 2>&1)
f=$(find /tmp/prof_r -name "*kernel_stats.csv" | head -1)
head -12 $f | cut -c1-200 | tee gpurun_out/r05q/render_kernel_stats.txt
tail -3 /tmp/prof_r.log
