#!/bin/bash
# round 6, GPU session 46: what stands between the update and the next lookup (11.8 us in the timeline, for BOTH chains) -- the update's
# dirty lines written back at the boundary?  AdamW's parameter stores / gradient zero-stores as non-temporal stores (the moments already are):
# -DPVD_ADAMW_NT_P / _G / both, three alternations, then the timeline of the best.
OUT=gpurun_out/r06s46
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
: > $OUT/ab.txt
for i in 1 2 3; do
  for v in base ntp ntg ntpg; do
    lib=$GRAFT_REPO_ROOT/aaai2023-pvd_amd/libpvd_hip.so; [ $v != base ] && lib=$GRAFT_REPO_ROOT/aaai2023-pvd_amd/libpvd_hip_$v.so
    PVD_HIP_LIB=$lib timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-psnr --sustained-steps 0 2>/dev/null | grep '^{' | tail -1 > /tmp/l.json
    python - "$v" "$i" <<'PY' >> $OUT/ab.txt
import json, sys
d = json.load(open("/tmp/l.json"))
print("%-5s run %s: %.4f ms/step" % (sys.argv[1], sys.argv[2], d["ms_per_step"]))
PY
  done
done
cat $OUT/ab.txt
(cd /tmp && rm -rf /tmp/prof_p && PVD_HIP_LIB=$GRAFT_REPO_ROOT/aaai2023-pvd_amd/libpvd_hip_ntpg.so timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --sustained-steps 0 > /tmp/prof_p.log 2>&1)
T=$(find /tmp/prof_p -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $T "k_vm_bwd_split" 22 2>&1 | tail -16 | cut -c1-100 | tee $OUT/timeline_ntpg.txt
true
