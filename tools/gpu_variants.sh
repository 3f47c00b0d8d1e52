#!/bin/bash
# GPU side: bench.py --config <cfg> under several environment variants.  tools/gpu_variants.sh <cfg> "A=1 B=2" "A=0" ...
cd $GRAFT_REPO_ROOT
CFG=$1; shift
for v in "$@"; do
  for rep in 1 2; do
  env $v timeout 400 python bench.py --config $CFG --no_cpu_baseline --no_other_configs --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg $CFG [$v]', round(d['ms_per_step'],3), d.get('kernel_ms'))"
  done
done
