#!/bin/bash
# Round 5 profiles (on the GPU box): profiles/r5_* of profiles/README.md.
#   1. scripts/profile_bench.sh r5 --in-flight 1 : the four passes (kernel trace, FETCH_SIZE, WRITE_SIZE, MFMA counters) over the ONE-STREAM
#      loop - every kernel's own duration and counters, nothing beside it;
#   2. r5d: a kernel trace of the DEFAULT command (two steps in flight): the line the driver measures, kernels time-sharing the chip;
#   3. r5_sc2pcr: a kernel trace of the SC2-PCR back-end on nuScenes-shaped pairs (configs[4] on one GPU).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
bash scripts/profile_bench.sh r5 --in-flight 1
export EYOC_BENCH_PAIR_CACHE=/tmp/eyoc_bench_pairs.pkl
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5d_trace -o r5d -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r5d_trace.log 2>&1
grep -h "^{" gpurun_out/r5d_trace.log | tail -1 | cut -c1-200
export EYOC_BENCH_PAIR_CACHE=/tmp/eyoc_bench_pairs_nus.pkl
python $R/bench.py --sc2pcr --nuscenes --pairs 16 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5_sc2pcr_trace -o r5_sc2pcr -- python $R/bench.py --sc2pcr --nuscenes --pairs 16 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --in-flight 1 > gpurun_out/r5_sc2pcr_trace.log 2>&1
grep -h "^{" gpurun_out/r5_sc2pcr_trace.log | tail -1 | cut -c1-200
ls gpurun_out/r5d_trace gpurun_out/r5_sc2pcr_trace
