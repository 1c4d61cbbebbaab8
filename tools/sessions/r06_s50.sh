#!/bin/bash
# round 6, GPU session 50: the 8-GPU configurations' students with EIGHT ranks sharing the one GPU over gloo (plumbing of the world-8 paths: the
# hash student's half-precision table exchange, the Plenoxel student's compact exchange); not a scaling figure.
OUT=gpurun_out/r06s50
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
: > $OUT/ranks8.txt
run() {
  tag=$1; shift
  s=$(date +%s)
  PVD_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 20 --warmup 5 --rays 1024 --no-cpu-baseline --no-psnr --sustained-steps 0 "$@" > $OUT/line_$tag.json 2> $OUT/err_$tag.txt; rc=$?
  e=$(date +%s)
  python - "$tag" "$rc" "$((e - s))" "$OUT/line_$tag.json" <<'PY' | tee -a $OUT/ranks8.txt
import json, sys
tag, rc, wall, path = sys.argv[1:]
lines = [l for l in open(path) if l.startswith("{")]
if rc != "0" or len(lines) != 1:
    print("%s: rc=%s wall=%ss JSON lines=%d" % (tag, rc, wall, len(lines)))
else:
    d = json.loads(lines[0])
    print("%s: rc=0 wall=%ss ONE line n_gpus=%d loss=%.3f exchange=%s" % (tag, wall, d["n_gpus"], d["config"]["loss"], d["config"].get("exchange", "")[:140]))
PY
}
run hash --student hash --teacher-pretrain 50
run tensors --teacher mlp --student tensors --data-type llff --teacher-pretrain 0
true
