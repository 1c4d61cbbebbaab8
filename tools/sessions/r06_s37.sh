#!/bin/bash
# round 6, GPU session 37: the deferred part of the update between the march and the teacher's lookup (PVD_PART_A_POS=mid) under ray-DP:
# one-rank RCCL step (collectives recorded into the graph), default exchange form, three alternations.
OUT=gpurun_out/r06s37
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
: > $OUT/ab.txt
for i in 1 2 3; do
  for v in end mid; do
    PVD_PART_A_POS=$v PVD_DP_FORCE=1 PVD_DP_PIPELINE=2 timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-psnr --sustained-steps 0 2>/dev/null | grep '^{' | tail -1 > /tmp/l.json
    python - "$v" "$i" <<'PY' >> $OUT/ab.txt
import json, sys
d = json.load(open("/tmp/l.json"))
print("one-rank RCCL  %-4s run %s: %.4f ms/step   %s" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["config"]["launch"][:70]))
PY
  done
done
cat $OUT/ab.txt
true
