#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
M=460000 timeout 120 python tools/bench_mlp_fused.py 2>&1 | grep fused
