#!/bin/bash
# Launches folded away around the update of a multi-step graph: in-kernel AdamW tail (last workgroup to be done), zero-after-read,
# the student's weight image packed on the forked branch -- against the separate launches.
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/${TAG:-r03f}; mkdir -p $OUT
if [ "${TESTS:-1}" = 1 ]; then timeout 900 python -m pytest tests/test_hip_fused_misc.py tests/test_hip_graph.py tests/test_hip_amp_parity.py tests/test_hip_dp_graph.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -5 | tee $OUT/pytest.log; fi
V0="PVD_ADAMW_TAIL_KERNEL=1 PVD_ADAMW_ZERO_IN_STEP=0 PVD_PACK_ON_BRANCH=0"
V1="PVD_ADAMW_TAIL_KERNEL=0 PVD_ADAMW_ZERO_IN_STEP=0 PVD_PACK_ON_BRANCH=0"
V2="PVD_ADAMW_TAIL_KERNEL=0 PVD_ADAMW_ZERO_IN_STEP=1 PVD_PACK_ON_BRANCH=0"
V3="PVD_ADAMW_TAIL_KERNEL=0 PVD_ADAMW_ZERO_IN_STEP=1 PVD_PACK_ON_BRANCH=1"
run() { env $1 timeout 300 python bench.py $2 --no-cpu-baseline 2>>$OUT/err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', '$2', 'ms/step %.4f' % d['ms_per_step'], 'loss %.4f psnr %.2f' % (d['config']['loss'], d['config']['psnr_student_vs_teacher_db']))" | tee -a $OUT/fold_launches_ab.txt; }
for v in "$V0" "$V1" "$V3" "$V0" "$V1" "$V3"; do run "$v" "--steps 200 --warmup 20"; done
for v in "$V0" "$V3"; do run "$v" "--steps 20 --warmup 5"; done
