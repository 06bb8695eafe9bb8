#!/bin/bash
# PMC passes over scripts/bench_spconv_layer.py (ONE=1: the level-1 64->64 layer); usage: scripts/pmc_layer.sh <tag>
tag=${1:-pl}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export ONE=1 PAIRS=16 WAVE=1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d gpurun_out/${tag}_sq -o p -- python scripts/bench_spconv_layer.py > gpurun_out/${tag}_sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/${tag}_tcc -o p -- python scripts/bench_spconv_layer.py > gpurun_out/${tag}_tcc.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d gpurun_out/${tag}_tcp -o p -- python scripts/bench_spconv_layer.py > gpurun_out/${tag}_tcp.log 2>&1
python - <<PY
import pandas as pd, glob
for sub in ("sq","tcc","tcp"):
    fs=glob.glob("gpurun_out/${tag}_%s/**/*counter_collection.csv"%sub, recursive=True)
    if not fs: print(sub,"no output"); continue
    df=pd.read_csv(fs[0]); df=df[df.Kernel_Name.str.contains("spconv")]
    print(df.groupby(["Kernel_Name","Counter_Name"]).Counter_Value.mean().unstack().T.to_string()[:3000])
PY
grep -h "Unable\|rror" gpurun_out/${tag}_*.log | head
