#!/bin/bash
# round 6, GPU session 26: the VM head's weight image packed on the lookup's forward launch (pvd_vm_forward_pack_rider): tests, then A/B
# (PVD_HEAD_PACK_RIDE was the A/B's own switch in pack_rides_on_lookup(); removed once the result was in: PVD_HEAD_DW_RIDE=0 turns both riders off)
OUT=gpurun_out/r06s26
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_hip_vm.py tests/test_hip_fused_misc.py tests/test_hip_graph.py tests/test_hip_golden_step.py tests/test_hip_fullsize.py -q -x 2>&1 | tail -5 | tee $OUT/tests.log
: > $OUT/ab.txt
for i in 1 2 3; do
  for v in 0 1; do
    PVD_HEAD_PACK_RIDE=$v timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-psnr --sustained-steps 0 2>/dev/null | grep '^{' | tail -1 > /tmp/l.json
    python - "$v" "$i" <<'PY' >> $OUT/ab.txt
import json, sys
d = json.load(open("/tmp/l.json"))
print("pack ride %s run %s: %.4f ms/step" % (sys.argv[1], sys.argv[2], d["ms_per_step"]))
PY
  done
done
cat $OUT/ab.txt
(cd /tmp && rm -rf /tmp/prof_p && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --sustained-steps 0 > /tmp/prof_p.log 2>&1)
T=$(find /tmp/prof_p -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $T "k_adamw(" 22 > $OUT/step_timeline.txt 2>&1; tail -18 $OUT/step_timeline.txt
true
