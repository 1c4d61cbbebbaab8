#!/bin/bash
# round 6, GPU session 7: targeted tests after the knob pruning / error-map wiring, the --error-map bench path
OUT=gpurun_out/r06s7
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests/test_hip_dp_graph.py tests/test_hip_dp_exchange.py tests/test_hip_infer_rounds.py tests/test_hip_vm.py tests/test_hip_bench_line.py tests/test_hip_fullsize.py -q 2>&1 | tail -8 | tee $OUT/tests.log
python tools/make_blender_scene.py /tmp/chair_scene --views 8 --res 64 > /dev/null 2>&1
timeout 600 python bench.py --workload teacher --steps 16 --warmup 32 --rays 1024 --no-cpu-baseline --data-root /tmp/chair_scene --error-map 2>$OUT/em.err | grep '^{' | tail -1 | cut -c1-900 | tee $OUT/bench_error_map.txt
tail -3 $OUT/em.err
true
