#!/bin/bash
# round 5: SC2-PCR path (nuScenes-shaped pairs, 16 per step): tests, the back-end alone, the bench step at one / two steps in flight, kernel statistics
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_sc2pcr.py -x -q 2>&1 | tail -3
PAIRS=16 python scripts/bench_sc2pcr.py
export EYOC_BENCH_PAIR_CACHE=/tmp/eyoc_bench_pairs_nus.pkl
for f in 1 2; do
python bench.py --sc2pcr --nuscenes --pairs 16 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --in-flight $f 2>gpurun_out/sc2_err_$f.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in-flight $f', round(d['value'],1),'pairs/s', round(d['ms_per_step'],2),'ms', d['stage_ms_per_step'], d['success_rate'])"
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_sc2 -o q -- python bench.py --sc2pcr --nuscenes --pairs 16 --steps 5 --warmup 1 --no-cpu-baseline --no-extras --in-flight 1 > gpurun_out/q_sc2.log 2>&1
python scripts/kstats.py gpurun_out/q_sc2 24
