#!/bin/bash
# round 6, GPU session 40: with the forked branch reordered, the launch shapes of its two heavy kernels again -- the deferred update's
# workgroup cap (512 shipped; -DPVD_A_BLOCKS=256 / 1024 / 2048) and the teacher lookup's (1024 = 4 per CU shipped; -DPVD_F_BLOCKS=512 / 768 / 2048).
OUT=gpurun_out/r06s40
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
: > $OUT/ab.txt
for i in 1 2 3; do
  for v in base a256 a1024 a2048 f512 f768 f2048; do
    lib=$GRAFT_REPO_ROOT/aaai2023-pvd_amd/libpvd_hip.so; [ $v != base ] && lib=$GRAFT_REPO_ROOT/aaai2023-pvd_amd/libpvd_hip_$v.so
    PVD_HIP_LIB=$lib timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-psnr --sustained-steps 0 2>/dev/null | grep '^{' | tail -1 > /tmp/l.json
    python - "$v" "$i" <<'PY' >> $OUT/ab.txt
import json, sys
d = json.load(open("/tmp/l.json"))
r = d["roofline"]
print("%-6s run %s: %.4f ms/step   lookup in step %.1f us" % (sys.argv[1], sys.argv[2], d["ms_per_step"], r["us_per_launch"]))
PY
  done
done
cat $OUT/ab.txt
true
