"""Turn the rocprofv3 outputs merged under gpurun_out/ into the small summaries committed under
profiles/ (the raw traces stay in scratch).  Usage: python scripts/summarize_profiles.py <tag>
expects gpurun_out/<tag>_trace, <tag>_fetch, <tag>_write (see profiles/README.md for the commands)."""
import json
import os
import re
import shutil
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)
g = os.path.join(ROOT, "gpurun_out")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("void ", "")
    return re.sub(r"\(.*", "", name)[:70]


shutil.copy(os.path.join(g, f"{tag}_trace", f"{tag}_kernel_stats.csv"), os.path.join(out, f"{tag}_kernel_stats.csv"))
bench = [l for l in open(os.path.join(g, f"{tag}_trace.log")) if l.startswith("{")][-1]
open(os.path.join(out, f"{tag}_bench_under_rocprof.json"), "w").write(bench)
# forwards under the profiler: timed + warm-up + the untimed allocator-settling steps of the pipelined loop (bench.py --overlap-maps)
steps = json.loads(bench)["steps"] + json.loads(bench)["warmup"] + json.loads(bench).get("config", {}).get("untimed_settle_steps", 0)

rows = {}
for kind, col in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    df = pd.read_csv(os.path.join(g, f"{tag}_{kind}", f"{tag}_counter_collection.csv"))
    df = df[df.Counter_Name == col]
    df["kernel"] = df["Kernel_Name"].map(short)
    for k, grp in df.groupby("kernel"):
        r = rows.setdefault(k, {"kernel": k, "launches": len(grp)})
        r[col + "_KB_total"] = float(grp.Counter_Value.sum())
t = pd.DataFrame(rows.values()).fillna(0.0)
# MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads
t["hbm_read_GB_per_step_x2corr"] = t["FETCH_SIZE_KB_total"] * 1024 * 2 / steps / 1e9
t["hbm_write_GB_per_step"] = t["WRITE_SIZE_KB_total"] * 1024 / steps / 1e9
t = t.sort_values("hbm_read_GB_per_step_x2corr", ascending=False)
t.to_csv(os.path.join(out, f"{tag}_hbm_traffic.csv"), index=False)
sp = t[t.kernel.str.startswith("spconv_") | t.kernel.str.startswith("tail_fused")]   # every sparse-conv launch (the fused 1x1 tail included)
summary = {"steps_profiled": steps,
           "spconv_read_GB_per_forward_x2corr": float(sp.hbm_read_GB_per_step_x2corr.sum()),
           "spconv_read_GB_per_forward_raw": float(sp.FETCH_SIZE_KB_total.sum() * 1024 / steps / 1e9),
           "spconv_write_GB_per_forward": float(sp.hbm_write_GB_per_step.sum()),
           "workload": json.loads(bench)["config"]["workload"]}
sys.path.insert(0, ROOT)
import bench as _bench  # noqa: E402
summary["csrc_sha16"] = _bench.csrc_sha16()          # bench.py quotes these counters only while the kernel sources are these
json.dump(summary, open(os.path.join(out, f"{tag}_spconv_traffic.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))

# optional 4th pass: matrix-pipe counters per kernel (gpurun_out/<tag>_mfma)
mf = os.path.join(g, f"{tag}_mfma", f"{tag}_counter_collection.csv")
if os.path.exists(mf):
    df = pd.read_csv(mf)
    df["kernel"] = df["Kernel_Name"].map(short)
    piv = df.pivot_table(index="kernel", columns="Counter_Name", values="Counter_Value", aggfunc="sum").fillna(0.0)
    piv["launches"] = df[df.Counter_Name == df.Counter_Name.iloc[0]].groupby("kernel").size()
    if "SQ_VALU_MFMA_BUSY_CYCLES" in piv and "SQ_BUSY_CYCLES" in piv:
        # MfmaUtil as rocprof-compute defines it: MFMA-busy cycles over (busy cycles x 4 SIMDs... per-SE counters are
        # already summed); reported as a plain ratio of the two counters
        piv["mfma_busy_over_sq_busy"] = piv["SQ_VALU_MFMA_BUSY_CYCLES"] / piv["SQ_BUSY_CYCLES"].clip(lower=1)
    piv = piv.sort_values(piv.columns[0], ascending=False)
    piv.to_csv(os.path.join(out, f"{tag}_mfma_counters.csv"))
    print(piv.head(12).to_string())
    if "SQ_INSTS_VALU_MFMA_MOPS_F16" in piv:
        # fp16 MFMA work the sparse-conv kernels ISSUE per forward (one MOP = 512 flop: 32 per v_mfma_f32_16x16x32_f16), zero
        # rows of partly empty 16-row blocks included - next to the useful products bench.py prices the roofline with
        sel = piv[[k.startswith("spconv_") or k.startswith("tail_fused") for k in piv.index]]
        summary["spconv_issued_fp16_mfma_flop_per_forward"] = float(sel["SQ_INSTS_VALU_MFMA_MOPS_F16"].sum()) * 512 / steps
        json.dump(summary, open(os.path.join(out, f"{tag}_spconv_traffic.json"), "w"), indent=1)
