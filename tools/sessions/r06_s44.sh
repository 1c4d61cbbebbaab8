#!/bin/bash
# round 6, GPU session 44: a hash student's table gradient under ray-DP as the half-precision table the scatter wrote (PVD_DP_HASH_WIRE=f16,
# new default) against widened into the fp32 bucket (f32, rounds 1-5): tests, then the one-rank RCCL step of hash->hash (configs[4]'s student).
OUT=gpurun_out/r06s44
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_dp_eval_occupancy.py tests/test_hip_bench_line.py -x -q -k "hash_student or eight_gpu" 2>&1 | grep -v Gloo | tail -5 | tee $OUT/tests.txt
: > $OUT/ab.txt
for i in 1 2 3; do
  for v in f32 f16; do
    PVD_DP_HASH_WIRE=$v PVD_DP_FORCE=1 PVD_DP_PIPELINE=2 timeout 300 python bench.py --student hash --steps 400 --warmup 40 --no-cpu-baseline --no-psnr --sustained-steps 0 --teacher-pretrain 100 2>$OUT/err_$v.txt | grep '^{' | tail -1 > /tmp/l.json
    python - "$v" "$i" <<'PY' >> $OUT/ab.txt
import json, sys
d = json.load(open("/tmp/l.json"))
print("hash->hash, one-rank RCCL, table gradient on the wire as %s  run %s: %.4f ms/step  loss %.4f  %s" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["config"]["loss"], d["config"].get("exchange", "")[:90]))
PY
  done
done
cat $OUT/ab.txt; tail -3 $OUT/err_f16.txt | cut -c1-300
true
