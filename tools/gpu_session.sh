#!/bin/bash
# One gpurun call of round 2 (edit per call).  Everything lands under gpurun_out/$TAG.
TAG=${TAG:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
(cd /tmp && rocprofv3 -L > "$GRAFT_REPO_ROOT/$OUT/counters.txt" 2>&1)
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
timeout 300 python tools/bench_grid.py > $OUT/bench_grid.log 2>&1
timeout 200 python tools/bench_grid_levels.py > $OUT/bench_grid_levels.log 2>&1
for lps in 0 2 4; do
  for c in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    n=$(echo $c | cut -d' ' -f1)
    (cd /tmp && PVD_GRID_LPS=$lps timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_${lps}_$n -- python "$GRAFT_REPO_ROOT/tools/pmc_grid_fwd.py" > /tmp/pmc_${lps}_$n.log 2>&1)
    f=$(find /tmp/pmc_${lps}_$n -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py $f | grep -i "grid_fwd" > $OUT/pmc_lps${lps}_$n.csv
    tail -3 /tmp/pmc_${lps}_$n.log > $OUT/pmc_lps${lps}_$n.log
  done
done
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
tail -5 $OUT/pytest.log; cat $OUT/bench_grid.log; cat $OUT/bench_grid_levels.log; cat $OUT/pmc_lps*.csv; cat $OUT/bench.json
