#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_pose.py tests/test_gpu_round2.py tests/test_gpu_ransac_scale.py -x -q 2>&1 | tail -3
export EYOC_BENCH_PAIR_CACHE=/tmp/eyoc_bench_pairs.pkl
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_serial -o q -- python bench.py --in-flight 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/q_serial.log 2>&1
python scripts/kstats.py gpurun_out/q_serial 60 | grep -E "k_generate|k_fit|k_count|k_bucket|k_rmse|k_select"
python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1),'pairs/s', round(d['ms_per_step'],2),'ms', d['undisturbed_pass']['stage_ms_per_step'], d['success_rate'])"
