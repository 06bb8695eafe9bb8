#!/bin/bash
# single-pair latency (scripts/bench_latency.py) with a kernel trace: which kernels and gaps a call is made of
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python scripts/bench_latency.py 2>&1 | tail -1
ITERS=10 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/lat_trace -o lat -- python scripts/bench_latency.py > gpurun_out/lat_trace.log 2>&1
tail -1 gpurun_out/lat_trace.log
ls gpurun_out/lat_trace
