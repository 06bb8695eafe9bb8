#!/bin/bash
# Round 6: per-kernel picture of the map build alone (128-cloud bench batch), lazy tables on / off.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for lazy in 1 0; do
  EYOC_MAPS_LAZY=$lazy timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6_maps_$lazy -o maps -- python scripts/bench_maps.py > gpurun_out/r6_maps_$lazy.log 2>&1
  tail -1 gpurun_out/r6_maps_$lazy.log
  python scripts/kstats.py gpurun_out/r6_maps_$lazy 24
done
