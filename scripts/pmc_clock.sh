#!/bin/bash
# effective clock of the staged layers: GRBM_GUI_ACTIVE / kernel duration (MI355X_MICROARCH.md "DVFS give-back")
tag=${1:-clk}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export ONLY_ST=1 PAIRS=${PAIRS:-16} ST_VARIANTS=${ST_VARIANTS:-1,128} ROUNDS=2
export EYOC_BENCH_PAIR_CACHE=/tmp/eyoc_staged_pairs.pkl
timeout 200 python scripts/bench_staged.py 2>&1 | tail -3
timeout 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/${tag} -o p -- python scripts/bench_staged.py > gpurun_out/${tag}.log 2>&1
python - <<PY
import pandas as pd, glob
fs=glob.glob("gpurun_out/${tag}/**/*counter_collection.csv", recursive=True)
df=pd.read_csv(fs[0]); print(list(df.columns))
df=df[df.Kernel_Name.str.contains("spconv_st")]
df["k"]=df.Kernel_Name.str.replace(r"void \(anonymous namespace\)::","",regex=True).str.slice(0,40)+" g"+df.Grid_Size.astype(str)
piv=df.pivot_table(index=["k","Dispatch_Id"],columns="Counter_Name",values="Counter_Value",aggfunc="sum").reset_index()
kt=glob.glob("gpurun_out/${tag}/**/*kernel_trace.csv", recursive=True)
if kt:
    t=pd.read_csv(kt[0]); t["dur"]=t.End_Timestamp-t.Start_Timestamp
    piv=piv.merge(t[["Dispatch_Id","dur"]],on="Dispatch_Id",how="left")
    piv["GHz"]=piv.GRBM_GUI_ACTIVE/piv.dur
pd.set_option("display.width",250)
print(piv.groupby("k").mean(numeric_only=True).to_string())
PY
tail -2 gpurun_out/${tag}.log
