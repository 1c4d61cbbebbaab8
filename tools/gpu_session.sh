#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 300 python bench.py --workload teacher 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['steps'], d['warmup'], d['ms_per_step'], d['config']['launch'][:60], d['roofline']['frac'])"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
