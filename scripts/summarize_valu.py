"""profiles/<tag>_valu_counters.csv and profiles/<tag>_valu.json from the `--pmc SQ_INSTS_VALU ...` passes of scripts/profile_valu.sh
(gpurun_out/<tag>_valu, gpurun_out/<tag>_sc2pcr_valu) and the kernel-trace summaries already under profiles/.  Run after
summarize_profiles.py:  python scripts/summarize_valu.py [tag = r5]"""
import csv
import json
import os
import re
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("void ", "")
    return re.sub(r"\(.*", "", name)[:60]


def kstats(path):
    return {short(r["Name"]): (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(path))}


TAG = sys.argv[1] if len(sys.argv) > 1 else "r5"
rates = {}
for line in open(os.path.join(ROOT, "profiles", "r5_valu_rates.txt")):
    m = re.match(r"(\S+)\s+blocks 1024: .*\((\d+\.\d+) G wave-inst/s chip\)", line)
    if m:
        rates[m.group(1)] = float(m.group(2))
res = {"csrc_sha16": bench.csrc_sha16(),
       "valu_issue_rates_G_wave_inst_per_s": {**{k: rates[k] for k in ("v_fma_f64", "v_fma_f32", "v_pk_fma_f32", "v_add_f32")},
                                              "source": "profiles/r5_valu_rates.txt (scripts/micro/valu_rates.hip, four waves per SIMD)"}}
for tag, key, names in ((TAG, "ransac", ["k_count", "k_generate<true>", "k_fit"]),
                        (TAG + "_sc2pcr", "sc2pcr", ["k_masks", "k_csr_fill", "k_nms", "k_seed_dense<32>", "k_seed_solve", "k_seed_fitness", "k_sc_spmv", "k_seed_topk<true>", "k_rank"])):
    df = pd.read_csv(os.path.join(ROOT, "gpurun_out", f"{tag}_valu", f"{tag}_counter_collection.csv"))
    df["kernel"] = df["Kernel_Name"].map(short)
    piv = df.pivot_table(index="kernel", columns="Counter_Name", values="Counter_Value", aggfunc="sum").fillna(0)
    piv["launches"] = df[df.Counter_Name == "SQ_INSTS_VALU"].groupby("kernel").size()
    piv.sort_values("SQ_INSTS_VALU", ascending=False).to_csv(os.path.join(ROOT, "profiles", f"{tag}_valu_counters.csv"))
    ks = kstats(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    d = {}
    for n in names:
        calls, tot = ks[n]
        inst = float(piv.loc[n, "SQ_INSTS_VALU"]) / float(piv.loc[n, "launches"])
        ms = tot / calls / 1e6
        d[n] = {"ms_per_launch": round(ms, 4), "valu_wave_instructions_per_launch": inst, "G_wave_inst_per_s": round(inst / ms / 1e6, 1)}
    res[key] = d
json.dump(res, open(os.path.join(ROOT, "profiles", f"{TAG}_valu.json"), "w"), indent=1)
print(json.dumps({k: {n: v["G_wave_inst_per_s"] for n, v in res[k].items()} for k in ("ransac", "sc2pcr")}, indent=1))
