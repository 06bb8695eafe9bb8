#!/bin/bash
# SC2-PCR per-seed stage: dense-block threshold x and the selection path, kernel times on the bench's nuScenes-shaped step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
export EYOC_BENCH_PAIR_CACHE=/tmp/eyoc_bench_pairs_nus.pkl
python bench.py --sc2pcr --nuscenes --pairs 16 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
for v in "2 1024"; do
  set -- $v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x_$1_$2 -o q -- python bench.py --sc2pcr --nuscenes --pairs 16 --steps 5 --warmup 1 --no-cpu-baseline --no-extras --in-flight 1 --sc2-dense-x $1 --sc2-list-cap $2 > gpurun_out/x.log 2>&1
  echo "== x=$1 cap=$2  $(grep -h '^{' gpurun_out/x.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'pairs/s')")"
  python scripts/kstats.py gpurun_out/x_$1_$2 30 | grep -E "k_seed_topk|k_seed_dense"
done
