#!/bin/bash
# round 5: one stream per step against two steps in flight (alternating runs on one box)
cd $GRAFT_REPO_ROOT
export EYOC_BENCH_PAIR_CACHE=/tmp/eyoc_bench_pairs.pkl
B="python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras"
f() { python -c "
import json,sys
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); print('$2', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],2), 'ms/step fwd', round(d['forward_ms_per_step'],2), 'conv', round(d['roofline']['ms_per_forward'],2), d['stage_ms_per_step'], 'succ', d['success_rate'], 'allocs', d['config'].get('device_allocs_in_timed_region'))
"; }
for rep in 1 2; do
  $B --in-flight 1 > gpurun_out/s_if1_$rep.log 2>&1; f gpurun_out/s_if1_$rep.log "in-flight 1          "
  $B --in-flight 2 > gpurun_out/s_if2_$rep.log 2>&1; f gpurun_out/s_if2_$rep.log "in-flight 2 start    "
  $B --in-flight 2 --maps-after layer:4 > gpurun_out/s_if2l4_$rep.log 2>&1; f gpurun_out/s_if2l4_$rep.log "in-flight 2 layer:4  "
  $B --in-flight 2 --maps-after layer:12 > gpurun_out/s_if2l12_$rep.log 2>&1; f gpurun_out/s_if2l12_$rep.log "in-flight 2 layer:12 "
done
