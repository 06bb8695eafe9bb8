#!/bin/bash
# quick kernel-trace passes: the serial loop (kernels undisturbed) and the SC2-PCR path on nuScenes-shaped pairs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
export EYOC_BENCH_PAIR_CACHE=/tmp/eyoc_bench_pairs.pkl
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_serial -o q -- python bench.py --in-flight 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/q_serial.log 2>&1
python scripts/kstats.py gpurun_out/q_serial 45
export EYOC_BENCH_PAIR_CACHE=/tmp/eyoc_bench_pairs_nus.pkl
python bench.py --sc2pcr --nuscenes --pairs 16 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_sc2 -o q -- python bench.py --sc2pcr --nuscenes --pairs 16 --steps 5 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/q_sc2.log 2>&1
grep -h "^{" gpurun_out/q_sc2.log | cut -c1-300
python scripts/kstats.py gpurun_out/q_sc2 40
