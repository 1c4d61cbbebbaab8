#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_fused_misc.py tests/test_hip_mlp_frozen.py tests/test_hip_workloads.py tests/test_hip_amp_parity.py -q -x 2>&1 | tail -4
timeout 300 python tools/bench_mlp_to_tensors.py 2>&1 | grep -v amdgpu | tail -1
timeout 300 python bench.py --student tensors --no-cpu-baseline --teacher-pretrain 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hash->tensors', d['ms_per_step'], d['config']['capture_fallback'], d['config']['loss'])"
timeout 120 python tools/bench_mlp_fused.py 2>&1 | grep fused
