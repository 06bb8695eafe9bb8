"""Timeline of ONE single-pair call from scripts/r6_latency_prof.sh's trace (kernels + copies, gaps between them).
usage: python scripts/lat_timeline.py [gpurun_out/lat_trace] [--full]"""
import sys
import pandas as pd
d = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "gpurun_out/lat_trace"
k = pd.read_csv(f"{d}/lat_kernel_trace.csv")
m = pd.read_csv(f"{d}/lat_memory_copy_trace.csv")
k["name"] = (k.Kernel_Name.str.replace("(anonymous namespace)::", "", regex=False).str.replace("void ", "", regex=False)
             .str.replace(r"\(.*", "", regex=True).str.replace(r"rocprim::ROCPRIM_\d+_NS::detail::", "rp::", regex=True).str[:48])
ev = pd.concat([k[["Start_Timestamp", "End_Timestamp", "name"]],
                m.assign(name="COPY " + m.Direction.astype(str))[["Start_Timestamp", "End_Timestamp", "name"]]]).sort_values("Start_Timestamp").reset_index(drop=True)
idx = [i for i, n in enumerate(ev.name) if n.startswith("k_select")]
a, b = idx[-2] + 3, idx[-1] + 3                     # a call = everything behind the previous call's two read-back copies
call = ev.iloc[a:b].copy()
t0 = ev.End_Timestamp.iloc[a - 1]                   # the previous call's last copy has landed
call["start_us"] = (call.Start_Timestamp - t0) / 1e3
call["dur_us"] = (call.End_Timestamp - call.Start_Timestamp) / 1e3
call["gap_us"] = (call.Start_Timestamp - call.End_Timestamp.shift(1).fillna(t0)) / 1e3
if "--full" in sys.argv:
    pd.set_option("display.width", 200); pd.set_option("display.max_rows", 300)
    print(call[["start_us", "dur_us", "gap_us", "name"]].to_string())
span = (call.End_Timestamp.max() - t0) / 1e3
print(f"one call: {len(call)} launches + copies, span {span:.0f} us, busy {call.dur_us.sum():.0f} us, idle {span - call.dur_us.sum():.0f} us")
big = call[call.gap_us > 12][["start_us", "gap_us", "name"]]
print("gaps > 12 us (in front of):"); print(big.to_string())
def grp(n):
    if n.startswith(("spconv", "eyoc::conv1", "tail_fused", "eyoc::k_permute")): return "forward"
    if n.startswith(("knn", "k_gather_rows", "k_gather_targets")): return "gather / NN"
    if n.startswith(("k_generate", "k_fit", "k_bucket", "k_count", "k_rmse", "k_select")) and not n.startswith("k_count_levels"): return "RANSAC"
    if n.startswith("COPY") or n.startswith("__amd"): return "copies / fills"
    return "map build"
print(call.assign(g=call.name.map(grp)).groupby("g").dur_us.agg(["sum", "count"]).round(1).to_string())
