#!/bin/bash
# round 6, GPU session 4: SURVEY 8(d)'s sweeps (occupancy 1 / 5 / 15 %, encoder micro-benchmarks), the split map over levels 7-13 with
# its staging, the 15 % scene against the oracle, roofline.sol in the line, the CPU baseline's thread scaling
OUT=gpurun_out/r06s4
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
line() { grep '^{' "$1" | tail -1; }
timeout 900 python -m pytest tests/test_hip_fullsize.py -q -k "15_percent" 2>&1 | tail -5 | tee $OUT/t_fullsize15.log
timeout 600 python tools/hash_sol.py 2>&1 | grep -v amdgpu | tee $OUT/hash_sol_table.txt
timeout 900 python tools/grid_microbench.py 2>&1 | grep -v amdgpu | tee $OUT/grid_microbench.txt
for occ in 1 5 15; do
  timeout 600 python bench.py --occupancy $occ --no-cpu-baseline --no-psnr > $OUT/bench_occ$occ.txt 2>> $OUT/bench.err
  line $OUT/bench_occ$occ.txt > $OUT/bench_occ$occ.json
  python - <<PY | tee -a $OUT/occupancy_sweep.txt
import json
d = json.load(open("$OUT/bench_occ$occ.json")); c = d["config"]; r = d["roofline"]
print("occupancy %2d: %.1f%% occupied, %.1f samples/ray, %d rows/step | %.4f ms/step = %.2f M rays/s = %.0f M samples/s | lookup in step %.1f us (frac %.3f), alone %.1f us (frac %.3f), sol %.1f us (frac %.3f)" % (
    $occ, 100 * c["occupied_fraction"], c["samples_per_ray"], c["padded_rows_per_step"], d["ms_per_step"], d["value"] / 1e6, c["samples_per_s"] / 1e6,
    r["us_per_launch"], r["frac"], r["alone"]["us_per_launch"], r["alone"]["frac"], (r.get("sol") or {}).get("us_per_launch", float("nan")), (r.get("sol") or {}).get("frac", float("nan"))))
PY
done
timeout 900 python tools/cpu_baseline_scaling.py --counts 1,8,16,32,64,128 --steps 2 2>&1 | grep -v amdgpu | tee $OUT/cpu_baseline_scaling.txt
tail -3 $OUT/bench.err
true
