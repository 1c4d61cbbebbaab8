#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o s -- python "$GRAFT_REPO_ROOT/bench.py" --student mlp --no-cpu-baseline --teacher-pretrain 50 --steps 40 --warmup 10 > /tmp/prof_s.log 2>&1)
python tools/step_timeline.py $(find /tmp/prof_s -name "*kernel_trace.csv" | head -1) "k_adamw(" 12 > /tmp/tl.txt; tail -1 /tmp/tl.txt
python - <<'PY'
import re,collections
acc=collections.defaultdict(lambda:[0,0.0])
for l in open('/tmp/tl.txt'):
    m=re.match(r"\s*\d+\s+[\d.]+\s+([\d.]+)\s+[-\d.]+\s+(.*)",l)
    if m:
        name=m.group(2)[:110]; acc[name][0]+=1; acc[name][1]+=float(m.group(1))
for k,(n,t) in sorted(acc.items(), key=lambda kv:-kv[1][1])[:16]:
    print("%8.1f us %4d x  %s"%(t,n,k))
PY
