#!/bin/bash
# round 6, GPU session 41: three hardware queues (never tried: round 2 measured 1 / 2 / 4 = runtime default / 8) against the shipped two,
# six processes each (the default of four had one slow process in ~15-25): bench.py --steps 400 --warmup 40.
OUT=gpurun_out/r06s41
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
: > $OUT/ab.txt
for i in 1 2 3 4 5 6; do
  for q in 2 3; do
    GPU_MAX_HW_QUEUES=$q PVD_FORKED_GRAPHS=1 timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-psnr --sustained-steps 0 2>/dev/null | grep '^{' | tail -1 > /tmp/l.json
    python - "$q" "$i" <<'PY' >> $OUT/ab.txt
import json, sys
d = json.load(open("/tmp/l.json"))
print("queues %s run %s: %.4f ms/step   %s" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["config"]["launch"][:60]))
PY
  done
done
cat $OUT/ab.txt
true
