#!/bin/bash
# round 6, GPU session 32: bench.py --gpus 4 / 8 started the way the driver starts --gpus 1 (it spawns its ranks itself), the ranks
# SHARING the one GPU over gloo (PVD_DIST_BACKEND=gloo): not a performance figure -- the world-4 / world-8 code paths (exchange layout
# in 4 / 8 chunks, padded tails, segmented capture, MAX over ranks, ONE line from rank 0) run end to end, in both exchange forms.
OUT=gpurun_out/r06s32
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
: > $OUT/ranks_sharing_one_gpu.txt
for n in 4 8; do
  for form in allreduce sharded; do
    s=$(date +%s)
    PVD_DIST_BACKEND=gloo PVD_DP_EXCHANGE=$form timeout 900 python bench.py --gpus $n --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --sustained-steps 0 --teacher-pretrain 50 > $OUT/line_${n}_$form.json 2> $OUT/err_${n}_$form.txt; rc=$?
    e=$(date +%s)
    python - "$n" "$form" "$rc" "$((e - s))" "$OUT/line_${n}_$form.json" <<'PY' | tee -a $OUT/ranks_sharing_one_gpu.txt
import json, sys
n, form, rc, wall, path = sys.argv[1:]
lines = [l for l in open(path) if l.startswith("{")]
if rc != "0" or len(lines) != 1:
    print("gpus %s %-9s rc=%s wall=%ss JSON lines=%d" % (n, form, rc, wall, len(lines)))
else:
    d = json.loads(lines[0])
    print("gpus %s %-9s rc=0 wall=%ss ONE line: n_gpus=%d value=%.3g rays/s (ranks share one GPU: not a scaling figure) ms_per_step=%.3f launch=%s exchange=%s" % (
        n, form, wall, d["n_gpus"], d["value"], d["ms_per_step"], d["config"].get("launch", "")[:60], d["config"].get("exchange", d["config"].get("dp_exchange", ""))))
PY
    tail -3 $OUT/err_${n}_$form.txt | cut -c1-300
  done
done
true
