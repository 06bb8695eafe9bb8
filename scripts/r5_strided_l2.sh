#!/bin/bash
# L2 -> CU traffic of the strided (gathering) convolutions: requests the vector L1s send to L2 per launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
export EYOC_BENCH_PAIR_CACHE=/tmp/eyoc_bench_pairs.pkl
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex "spconv_wave_kernel|spconv_st_asm_kernel<64" --output-format csv -d gpurun_out/q_l2 -o q -- python bench.py --in-flight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/q_l2.log 2>&1
  grep -h "Unable to find" gpurun_out/q_l2.log | head -3
  python - <<PY
import pandas as pd, re
try:
    df=pd.read_csv("gpurun_out/q_l2/q_counter_collection.csv")
    df["k"]=df.Kernel_Name.map(lambda s: re.sub(r"\(.*","",s.replace("void (anonymous namespace)::","")))
    g=df.groupby(["k","Counter_Name"]).Counter_Value.agg(["sum","count"])
    g["per_launch"]=g["sum"]/g["count"]
    print(g[["per_launch","count"]].to_string())
except Exception as e: print("no data", e)
PY
done
