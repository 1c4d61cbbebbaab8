#!/bin/bash
# round 6, GPU session 35: the library built for --offload-arch=gfx950:xnack- (code that need not be replayable after a page fault: the
# compiler may release address registers early) against the default "xnack any" build; bench.py --steps 400 --warmup 40, three alternations,
# then the kernel-level parity files on the variant.
OUT=gpurun_out/r06s35
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
: > $OUT/ab.txt
for i in 1 2 3; do
  for v in base xn; do
    lib=$GRAFT_REPO_ROOT/aaai2023-pvd_amd/libpvd_hip.so; [ $v != base ] && lib=$GRAFT_REPO_ROOT/aaai2023-pvd_amd/libpvd_hip_$v.so
    PVD_HIP_LIB=$lib timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-psnr --sustained-steps 0 2>/dev/null | grep '^{' | tail -1 > /tmp/l.json
    python - "$v" "$i" <<'PY' >> $OUT/ab.txt
import json, sys
d = json.load(open("/tmp/l.json"))
r = d["roofline"]
print("%-5s run %s: %.4f ms/step   lookup in step %.1f us  alone %s" % (sys.argv[1], sys.argv[2], d["ms_per_step"], r["us_per_launch"], r.get("alone", {}).get("us_per_launch") if isinstance(r.get("alone"), dict) else r.get("alone")))
PY
  done
done
cat $OUT/ab.txt
PVD_HIP_LIB=$GRAFT_REPO_ROOT/aaai2023-pvd_amd/libpvd_hip_xn.so timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_vm.py tests/test_hip_head.py -x -q 2>&1 | tail -3 | tee -a $OUT/ab.txt
true
