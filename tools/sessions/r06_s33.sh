#!/bin/bash
# round 6, GPU session 33: A/B of the run length a wave walks in the VM lookup (-DPVD_VM_FWD_CHUNK=8 / 32 against 16; backward
# -DPVD_VM_BWD_CHUNK=32 / 48 against 64): shorter serial chains per wave against more window reloads / fewer merged atomics.
OUT=gpurun_out/r06s33
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
: > $OUT/ab.txt
for i in 1 2 3; do
  for v in base vmf8 vmf32 vmb32 vmb48; do
    lib=$GRAFT_REPO_ROOT/aaai2023-pvd_amd/libpvd_hip.so; [ $v != base ] && lib=$GRAFT_REPO_ROOT/aaai2023-pvd_amd/libpvd_hip_$v.so
    PVD_HIP_LIB=$lib timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-psnr --sustained-steps 0 2>/dev/null | grep '^{' | tail -1 > /tmp/l.json
    python - "$v" "$i" <<'PY' >> $OUT/ab.txt
import json, sys
d = json.load(open("/tmp/l.json"))
print("%-6s run %s: %.4f ms/step   loss %.4f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["config"]["loss"]))
PY
  done
done
cat $OUT/ab.txt
true
