#!/bin/bash
# timing-only ablations of conv1_bf_kernel: 1 no weight loads for the LDS table, 2 class weights without the window table, 3 class weights
# from 8 LDS rows, 4 one chunk per class only
cd $GRAFT_REPO_ROOT
export EYOC_BENCH_PAIR_CACHE=/tmp/eyoc_bench_pairs.pkl
for v in 0 1 2 3 4; do
  lib=$GRAFT_REPO_ROOT/eyoc_amd/lib/libeyoc_hip_c1abl$v.so; [ $v = 0 ] && lib=$GRAFT_REPO_ROOT/eyoc_amd/lib/libeyoc_hip.so
  EYOC_HIP_LIB=$lib python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extras --in-flight 1 --conv1-kernel 3 --inlier-ratio 0 --verbose 2>&1 | grep -E "^conv1 " | sed "s/^/abl $v: /"
done
