cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do for v in 0 1; do
PVD_X_JOIN2=$v timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-psnr --sustained-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('TIMING PROBE (not a valid step when 1) join every second step=$v run $i: %.4f ms/step' % d['ms_per_step'])"
done; done
