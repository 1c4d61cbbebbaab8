#!/bin/bash
TAG=${TAG:-r02l}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_dp_graph.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
for mode in "single" "force_ingraph" "force_segmented"; do
  case $mode in
    single) env="";;
    force_ingraph) env="PVD_DP_FORCE=1 PVD_DP_INGRAPH=1";;
    force_segmented) env="PVD_DP_FORCE=1 PVD_DP_INGRAPH=0";;
  esac
  env $env timeout 300 python bench.py --no-cpu-baseline --teacher-pretrain 100 > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err
  echo "== $mode: $(python -c 'import sys,json; d=json.loads(open(sys.argv[1]).read()); print(d["ms_per_step"], d["config"].get("exchange"), d["config"]["launch"])' $OUT/bench_$mode.json 2>&1 | tail -1)"
done
