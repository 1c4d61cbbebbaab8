#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/bench_mlp_to_tensors.py 2>&1 | grep -v amdgpu | tail -1
timeout 600 python -m pytest tests/test_hip_head.py tests/test_hip_workloads.py tests/test_hip_infer_rounds.py -q -x 2>&1 | tail -2
