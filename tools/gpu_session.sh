#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4; do timeout 600 python -m pytest tests/test_hip_fused_misc.py tests/test_hip_dp_graph.py -q -x 2>&1 | grep -E "^E  |passed|failed" | head -6; done
