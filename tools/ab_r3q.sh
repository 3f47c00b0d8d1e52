#!/bin/bash
cd $GRAFT_REPO_ROOT
for st in 5 20 40 20; do
timeout 600 python bench.py --config 5b --no_cpu_baseline --steps $st --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg5b steps $st', d['ms_per_step'])"
done
