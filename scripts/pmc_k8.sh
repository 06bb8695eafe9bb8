#!/bin/bash
# HBM-side reads of the K = 8 (strided / transposed) layers, standalone: scripts/pmc_k8.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export K8=1 PAIRS=16
for mode in 0 2; do
  RS=$mode timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/k8_$mode -o p -- python scripts/bench_spconv_layer.py > gpurun_out/k8_$mode.log 2>&1
  grep "lvl" gpurun_out/k8_$mode.log
  python - <<PY
import pandas as pd, glob
fs=glob.glob("gpurun_out/k8_$mode/**/*counter_collection.csv", recursive=True)
df=pd.read_csv(fs[0]); df=df[df.Kernel_Name.str.contains("spconv_")]
df["k"]=df.Kernel_Name.str.replace(r"\(anonymous namespace\)::","",regex=True).str.replace(r"\(.*","",regex=True).str[:44]
t=df.pivot_table(index=["Dispatch_Id","k","Grid_Size"],columns="Counter_Name",values="Counter_Value",aggfunc="sum")
print(t.groupby(["k","Grid_Size"]).mean().to_string())
PY
done
