#!/bin/bash
# On the GPU box: the rocprofv3 passes behind profiles/<tag>_* (see profiles/README.md).  --pmc passes never share a
# run with a trace domain.   usage: scripts/profile_bench.sh <tag> [extra bench args]
tag=${1:-r2}; shift || true
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras $*"
cd $R
# the synthetic pairs are generated ONCE, outside the profiler (a forked worker pool under rocprofv3's signal handlers
# has hung a pass), and every pass is bounded
export EYOC_BENCH_PAIR_CACHE=/tmp/eyoc_bench_pairs.pkl
python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras $* > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -o $tag -- $B > gpurun_out/${tag}_trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -o $tag -- $B > gpurun_out/${tag}_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -o $tag -- $B > gpurun_out/${tag}_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d gpurun_out/${tag}_mfma -o $tag -- $B > gpurun_out/${tag}_mfma.log 2>&1
grep -h "^{" gpurun_out/${tag}_trace.log | tail -1 | cut -c1-400
ls gpurun_out/${tag}_*/ | head -30
grep -h "Unable to find\|rror" gpurun_out/${tag}_mfma.log | head -5
