#!/bin/bash
TAG=${TAG:-r02n}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_head.py tests/test_hip_vm.py tests/test_hip_infer_rounds.py tests/test_hip_render_parity.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -v "^$" $OUT/pytest.log | grep -v "^  \|^    " | tail -40
timeout 300 python tools/bench_render.py > $OUT/bench_render.log 2>&1; grep -v amdgpu.ids $OUT/bench_render.log | tail -20
