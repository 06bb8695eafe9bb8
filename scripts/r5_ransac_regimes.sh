#!/bin/bash
# the registration stage at higher planted inlier ratios (more survivors per pair: the count dominates)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for p in 0.3 0.45 0.6; do
  python bench.py --inlier-ratio $p --steps 4 --warmup 1 --in-flight 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('p=$p', round(d['value'],1),'pairs/s', round(d['ms_per_step'],2),'ms', d['stage_ms_per_step'], 'survivors/pair', d.get('survivors_per_pair'), d['success_rate'])"
done
