#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_head.py -q 2>&1 | grep -E "^E  |assert|FAILED|passed|failed" | head -40
