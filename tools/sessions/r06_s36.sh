#!/bin/bash
# round 6, GPU session 36: A/B of WHERE on the forked branch the deferred part of the update (AdamW part A, HBM streaming) sits: at the end
# (shipped: batch, march, teacher lookup + head, its compositing, part A) or between the march and the teacher's lookup (PVD_PART_A_POS=mid:
# part A next to the student's head backward, the lookup next to the table scatter).  No extra dependency edge either way.
OUT=gpurun_out/r06s36
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
: > $OUT/ab.txt
for i in 1 2 3; do
  for v in end mid start; do
    PVD_PART_A_POS=$v timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-psnr --sustained-steps 0 2>/dev/null | grep '^{' | tail -1 > /tmp/l.json
    python - "$v" "$i" <<'PY' >> $OUT/ab.txt
import json, sys
d = json.load(open("/tmp/l.json"))
r = d["roofline"]
print("%-4s run %s: %.4f ms/step   lookup in step %.1f us   loss %.4f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], r["us_per_launch"], d["config"]["loss"]))
PY
  done
done
cat $OUT/ab.txt
(cd /tmp && rm -rf /tmp/prof_p && PVD_PART_A_POS=mid timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --sustained-steps 0 > /tmp/prof_p.log 2>&1)
T=$(find /tmp/prof_p -name "*kernel_trace.csv" | head -1)
: > $OUT/kernel_populations_mid.txt
for k in k_hash_fwd_fused k_vm_bwd_split k_vm_fwd "k_adamw(" k_head_bwd k_head_fwd k_composite_bwd_wave k_composite_fwd_wave k_march_count_wave; do python tools/kernel_populations.py $T "$k" >> $OUT/kernel_populations_mid.txt; done
grep "sharing" $OUT/kernel_populations_mid.txt
python tools/step_timeline.py $T "k_vm_bwd_split" 22 > $OUT/step_timeline_mid.txt 2>&1; tail -20 $OUT/step_timeline_mid.txt
true
