#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02u
timeout 300 python tools/probe_fused_in_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02u/probe.log
