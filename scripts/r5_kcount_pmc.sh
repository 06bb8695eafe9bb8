cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
export EYOC_BENCH_PAIR_CACHE=/tmp/eyoc_bench_pairs.pkl

python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex "k_count" --output-format csv -d gpurun_out/q_kcpmc$i -o q -- python bench.py --in-flight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/q_kcpmc$i.log 2>&1
  python - <<PY
import pandas as pd
df=pd.read_csv("gpurun_out/q_kcpmc$i/q_counter_collection.csv")
df=df[df.Kernel_Name.str.contains("k_count\(")]
print(df.groupby("Counter_Name").Counter_Value.sum()/df[df.Counter_Name==df.Counter_Name.iloc[0]].shape[0])
PY
done
