#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_hip_edges.py -q -x 2>&1 | tail -6
