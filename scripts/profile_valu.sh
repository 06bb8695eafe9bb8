#!/bin/bash
# VALU issue counters per kernel (RANSAC's k_count / k_generate, SC2-PCR's kernels): usage scripts/profile_valu.sh <tag> <bench args...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d gpurun_out/${tag}_valu -o $tag -- python $R/bench.py --no-cpu-baseline --no-extras --in-flight 1 $* > gpurun_out/${tag}_valu.log 2>&1
grep -h "Unable to find\|rror" gpurun_out/${tag}_valu.log | head -3
ls gpurun_out/${tag}_valu
