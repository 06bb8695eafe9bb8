cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_sc2pcr.py -x -q -m gpu 2>&1 | tail -5
export EYOC_BENCH_PAIR_CACHE=/tmp/eyoc_bench_pairs_nus.pkl
python $R/bench.py --sc2pcr --nuscenes --pairs ${PAIRS:-16} --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
python $R/bench.py --sc2pcr --nuscenes --pairs ${PAIRS:-16} --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | grep "^{" | cut -c1-260
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sc2x_trace -o sc2x -- python $R/bench.py --sc2pcr --nuscenes --pairs ${PAIRS:-16} --steps 10 --warmup 2 --no-cpu-baseline --no-extras --in-flight 1 > gpurun_out/sc2x_trace.log 2>&1
python - <<PY
import pandas as pd, glob
f=glob.glob("gpurun_out/sc2x_trace/**/sc2x_kernel_stats.csv", recursive=True)[0]
df=pd.read_csv(f)
df["k"]=df.Name.str.replace("(anonymous namespace)::","",regex=False).str.replace("void ","",regex=False).str.replace(r"\(.*","",regex=True)
sc=df[df.k.str.contains("k_sc_|k_csr|k_seed|k_masks|k_nms|k_rank|k_seeds|k_refine|k_init")]
print(sc[["k","Calls","TotalDurationNs","AverageNs"]].to_string())
print(sc[sc.k=="k_sc_spmv"][["MinNs","MaxNs"]].to_string() if "MinNs" in sc else ""); print("back-end total per step (ms):", sc.TotalDurationNs.sum()/1e6/18)
PY
