#!/bin/bash
# Fresh-process loops over a set of GPU test files, keeping the FULL log of every run that does not end green.
#   tools/flake_hunt.sh <tag> <loops> <pytest args...>      (env of the caller = the variant under test)
# Writes gpurun_out/<tag>/summary.txt (one line per run: rc, seconds, pytest's last line) and run_<i>.log for failures.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
tag=$1; loops=$2; shift 2
out=gpurun_out/$tag; mkdir -p "$out"
: > "$out/summary.txt"
for i in $(seq 1 "$loops"); do
  t0=$(date +%s.%N)
  timeout 600 python -X faulthandler -m pytest "$@" -x -q -m gpu -p no:cacheprovider > "$out/run_$i.log" 2>&1
  rc=$?
  t1=$(date +%s.%N)
  printf "run %02d rc=%d %.1fs %s\n" "$i" "$rc" "$(echo "$t1 - $t0" | bc)" "$(tail -1 "$out/run_$i.log" | cut -c1-160)" >> "$out/summary.txt"
  if [ "$rc" -eq 0 ]; then rm -f "$out/run_$i.log"; fi
done
echo "== $tag: $(grep -c 'rc=0 ' "$out/summary.txt")/$loops green"
