#!/bin/bash
# round 6, GPU session 39: SURVEY 8(d)'s occupancy sweep again on the final build (the forked branch reordered)
OUT=gpurun_out/r06s39
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
line() { grep '^{' "$1" | tail -1; }
: > $OUT/occupancy_sweep.txt
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
python - <<'PY'
import json
out = {}
for occ in (1, 5, 15):
    out[str(occ)] = json.load(open("gpurun_out/r06s39/bench_occ%d.json" % occ))
json.dump(out, open("gpurun_out/r06s39/occupancy_sweep.json", "w"))
PY
true
