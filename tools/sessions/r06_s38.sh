cd "$GRAFT_REPO_ROOT"
for i in 1 2; do for v in end mid; do for st in hash tensors; do
PVD_PART_A_POS=$v timeout 300 python bench.py --student $st --steps 400 --warmup 40 --no-cpu-baseline --no-psnr --sustained-steps 0 --teacher-pretrain 100 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$st $v $i %.4f' % d['ms_per_step'], d['config']['update'][:40])"
done; done; done
