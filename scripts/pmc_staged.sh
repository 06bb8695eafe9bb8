#!/bin/bash
# SQ / LDS counter passes over scripts/bench_staged.py (level-0 64->64 and level-2 128->128 staged layers), every
# variant of the staged kernel; usage: scripts/pmc_staged.sh <tag>     (--pmc passes only: never with a trace domain)
tag=${1:-ps}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export ONLY_ST=1 PAIRS=${PAIRS:-16} ST_VARIANTS=${ST_VARIANTS:-0,1,2}
# pairs generated ONCE outside the profiler (a forked worker pool under rocprofv3's signal handlers hangs); passes bounded
export EYOC_BENCH_PAIR_CACHE=/tmp/eyoc_staged_pairs.pkl
timeout 200 python scripts/bench_staged.py 2>&1 | tail -3
timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d gpurun_out/${tag}_sq -o p -- python scripts/bench_staged.py > gpurun_out/${tag}_sq.log 2>&1
timeout 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_MISC --output-format csv -d gpurun_out/${tag}_lds -o p -- python scripts/bench_staged.py > gpurun_out/${tag}_lds.log 2>&1
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_tr -o p -- python scripts/bench_staged.py > gpurun_out/${tag}_tr.log 2>&1
python - <<PY
import pandas as pd, glob
for sub in ("sq","lds"):
    fs=glob.glob("gpurun_out/${tag}_%s/**/*counter_collection.csv"%sub, recursive=True)
    if not fs: print(sub,"no output"); continue
    df=pd.read_csv(fs[0]); df=df[df.Kernel_Name.str.contains("spconv_st")]
    df["k"]=df.Kernel_Name.str.replace(r"void \(anonymous namespace\)::","",regex=True).str.slice(0,44)+" g"+df.Grid_Size.astype(str)
    pd.set_option("display.width",250)
    print(df.groupby(["k","Counter_Name"]).Counter_Value.mean().unstack().T.to_string())
PY
python scripts/kstats.py gpurun_out/${tag}_tr 12
grep -h "Unable\|rror" gpurun_out/${tag}_*.log | head
tail -3 gpurun_out/${tag}_tr.log
