cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/prof_p && PVD_FORKED_GRAPHS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --sustained-steps 0 > /tmp/prof_p.log 2>&1)
grep '^{' /tmp/prof_p.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no fork: %.4f ms/step' % d['ms_per_step'], d['config']['launch'][:80])"
T=$(find /tmp/prof_p -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $T "k_vm_bwd_split" 22 2>&1 | tail -18 | cut -c1-100
