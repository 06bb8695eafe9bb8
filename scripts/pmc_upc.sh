#!/bin/bash
# HBM traffic of the class-major transposed layers (scripts/bench_upc.py): separate --pmc passes, FETCH_SIZE / WRITE_SIZE in KB
tag=${1:-upc}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export ONLY_UPC=1 REPS=2
timeout 240 rocprofv3 --kernel-include-regex 'upc' --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_rd -o p -- python scripts/bench_upc.py > gpurun_out/${tag}_rd.log 2>&1
timeout 240 rocprofv3 --kernel-include-regex 'upc' --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_wr -o p -- python scripts/bench_upc.py > gpurun_out/${tag}_wr.log 2>&1
python - <<PY
import pandas as pd, glob
for sub in ("rd","wr"):
    fs=glob.glob("gpurun_out/${tag}_%s/**/*counter_collection.csv"%sub, recursive=True)
    if not fs: print(sub,"no output"); continue
    df=pd.read_csv(fs[0]); df=df[df.Kernel_Name.str.contains("spconv_upc_kernel|k_upc")]
    df["k"]=df.Kernel_Name.str.slice(0,60)+" grid "+df.Grid_Size.astype(str)
    print((df.groupby(["k","Counter_Name"]).Counter_Value.mean()/1e6).round(3).to_string())
PY
