#!/bin/bash
# N consecutive fresh-process runs of the driver's GPU test command; one summary line per run, full log kept for a run that is
# not green.   tools/flake_check.sh [N=10] [TAG=r03flake]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
N=${1:-10}; TAG=${2:-r03flake}; OUT=gpurun_out/$TAG; mkdir -p $OUT; : > $OUT/flake.txt
echo "command: python -m pytest tests/ -x -q -m gpu -p no:cacheprovider   (HEAD $(cat .git_head 2>/dev/null || echo '?'))" >> $OUT/flake.txt
for i in $(seq 1 $N); do
  s=$(date +%s)
  timeout 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/run_$i.log 2>&1; rc=$?
  e=$(date +%s)
  echo "run $i: rc=$rc wall=$((e - s))s  $(tail -1 $OUT/run_$i.log)" | tee -a $OUT/flake.txt
  [ $rc -eq 0 ] && rm -f $OUT/run_$i.log
done
echo "green: $(grep -c 'rc=0 ' $OUT/flake.txt) of $N" | tee -a $OUT/flake.txt
