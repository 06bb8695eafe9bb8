cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
export EYOC_BENCH_PAIR_CACHE=/tmp/eyoc_bench_pairs.pkl
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
for v in base w4 w5; do
  if [ $v != base ]; then export EYOC_HIP_LIB=$R/gpurun_tmp/libeyoc_$v.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_kc_$v -o q -- python bench.py --in-flight 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/q_kc_$v.log 2>&1
  echo $v; python scripts/kstats.py gpurun_out/q_kc_$v 60 | grep -E "k_count  |k_count "
done
