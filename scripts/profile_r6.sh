#!/bin/bash
# Round 6 profiles (on the GPU box): profiles/r6_* of profiles/README.md.
#   1. scripts/profile_bench.sh r6 --in-flight 1 : the four passes (kernel trace, FETCH_SIZE, WRITE_SIZE, MFMA counters) over the ONE-STREAM
#      loop - every kernel's own duration and counters, nothing beside it;
#   2. r6d: a kernel trace of the DEFAULT command (two steps in flight): the line the driver measures, kernels time-sharing the chip;
#   3. r6_sc2pcr: a kernel trace of the SC2-PCR back-end on nuScenes-shaped pairs (configs[4] on one GPU);
#   4. VALU issue counters (RANSAC, SC2-PCR kernels);
#   5. L1 -> L2 request counters of the strided convolutions, staged (default) and gathering (--down-kernel 0: the round-5 kernel), and
#      the SQ counter passes over RANSAC's k_count (the evidence EXPERIMENTS.md quotes, kept as text summaries)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
bash scripts/profile_bench.sh r6 --in-flight 1
export EYOC_BENCH_PAIR_CACHE=/tmp/eyoc_bench_pairs.pkl
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6d_trace -o r6d -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r6d_trace.log 2>&1
grep -h "^{" gpurun_out/r6d_trace.log | tail -1 | cut -c1-200
bash scripts/profile_valu.sh r6 --steps 3 --warmup 1
# strided-layer L2 requests: staged (default) and gathering
for dk in 1 0; do
  for set in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    timeout 300 rocprofv3 --pmc $set --kernel-include-regex "spconv_wave_kernel|spconv_st_asm_kernel<.*128" --output-format csv -d gpurun_out/r6_l2_$dk -o q -- python bench.py --in-flight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --down-kernel $dk > gpurun_out/r6_l2.log 2>&1
    python - <<PY >> gpurun_out/r6_strided_l2.txt
import pandas as pd, re
print("== down-kernel $dk ($( [ $dk = 1 ] && echo 'staged on 128-row tiles: conv2 / conv3 / conv4 + the two 256-channel stride-1 layers' || echo 'gathering kernel of rounds 1-5' )), counters: $set")
try:
    df=pd.read_csv("gpurun_out/r6_l2_$dk/q_counter_collection.csv")
    df["k"]=df.Kernel_Name.map(lambda s: re.sub(r"\(.*","",s.replace("void (anonymous namespace)::","")))
    g=df.groupby(["k","Counter_Name"]).Counter_Value.agg(["sum","count"])
    g["per_launch"]=g["sum"]/g["count"]
    print(g[["per_launch","count"]].to_string())
except Exception as e: print("no data", e)
PY
  done
done
# k_count SQ counters
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex "k_count" --output-format csv -d gpurun_out/r6_kcpmc$i -o q -- python bench.py --in-flight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r6_kcpmc$i.log 2>&1
  python - <<PY >> gpurun_out/r6_kcount_pmc.txt
import pandas as pd
df=pd.read_csv("gpurun_out/r6_kcpmc$i/q_counter_collection.csv")
df=df[df.Kernel_Name.str.contains("k_count\\\\(")]
print((df.groupby("Counter_Name").Counter_Value.sum()/df[df.Counter_Name==df.Counter_Name.iloc[0]].shape[0]).to_string())
PY
done
export EYOC_BENCH_PAIR_CACHE=/tmp/eyoc_bench_pairs_nus.pkl
python $R/bench.py --sc2pcr --nuscenes --pairs 16 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6_sc2pcr_trace -o r6_sc2pcr -- python $R/bench.py --sc2pcr --nuscenes --pairs 16 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --in-flight 1 > gpurun_out/r6_sc2pcr_trace.log 2>&1
grep -h "^{" gpurun_out/r6_sc2pcr_trace.log | tail -1 | cut -c1-200
bash scripts/profile_valu.sh r6_sc2pcr --sc2pcr --nuscenes --pairs 16 --steps 5 --warmup 1
ls gpurun_out/r6d_trace gpurun_out/r6_sc2pcr_trace
cat gpurun_out/r6_strided_l2.txt | head -60
cat gpurun_out/r6_kcount_pmc.txt
