#!/bin/bash
# round 6, GPU session 31: A/B of a march count pass confined to fewer resident wavefronts (-DPVD_MARCH_GROUPS=512 / 256: 2 / 1 waves
# per SIMD walking 2 / 4 rays each instead of 4 waves per SIMD with one ray each) -- does the student's head forward / compositing,
# which run next to it, get their wave slots back?  Three alternations of 400 steps each, then the in-step populations of the variants.
OUT=gpurun_out/r06s31
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
: > $OUT/ab.txt
for i in 1 2 3; do
  for v in base mg512 mg256; do
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
for v in mg512 mg256; do
(cd /tmp && rm -rf /tmp/prof_p && PVD_HIP_LIB=$GRAFT_REPO_ROOT/aaai2023-pvd_amd/libpvd_hip_$v.so timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --sustained-steps 0 > /tmp/prof_p.log 2>&1)
T=$(find /tmp/prof_p -name "*kernel_trace.csv" | head -1)
: > $OUT/kernel_populations_$v.txt
for k in k_hash_fwd_fused k_vm_bwd_split k_vm_fwd "k_adamw(" k_head_bwd k_head_fwd k_composite_bwd_wave k_composite_fwd_wave k_march_count_wave; do python tools/kernel_populations.py $T "$k" >> $OUT/kernel_populations_$v.txt; done
echo "== $v"; grep -E "k_head_fwd|k_composite_bwd|k_march_count|k_head_bwd" $OUT/kernel_populations_$v.txt
done
true
