cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4 5 6 7 8; do for v in 0 1; do
PVD_X_PRIO=$v timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-psnr --sustained-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('capture stream high priority=$v run $i: %.4f ms/step' % d['ms_per_step'])"
done; done
