"""Stage times of a batch of 8 pairs (BASELINE configs[2]) - diagnostics."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, eyoc_amd
from eyoc_amd.harness import DeviceBatch, RegistrationConfig, RegistrationPipeline
P = int(os.environ.get("PAIRS", "8"))
pairs = bench.make_pairs(list(range(P)))
dev = torch.device("cuda:0")
model, _sd = bench.build_model(dev, 0)
cfg = RegistrationConfig(ransac_max_iteration=4000000)
pipe = RegistrationPipeline(model, cfg)
batch = DeviceBatch(pairs, list(range(P)), dev, descriptor=dict(inlier_ratio=0.3))
if os.environ.get("ST_GROUP"):
    from eyoc_amd import _lib
    _lib.knob("eyoc_spconv_st_group_rows", int(os.environ["ST_GROUP"]))
if os.environ.get("EYOC_DOWN"):
    from eyoc_amd import _lib
    _lib.knob("eyoc_spconv_select_down_kernel", int(os.environ["EYOC_DOWN"]))
if os.environ.get("FUSE_TAIL"):
    from eyoc_amd import _lib
    _lib.knob("eyoc_model_fuse_tail", int(os.environ["FUSE_TAIL"]))
if os.environ.get("LAZY"):
    from eyoc_amd import _lib
    _lib.knob("eyoc_maps_lazy_tables", int(os.environ["LAZY"]))
if os.environ.get("RANSAC_PRUNE"):
    from eyoc_amd import _lib
    _lib.knob("eyoc_ransac_select_pruning", int(os.environ["RANSAC_PRUNE"]))
if os.environ.get("ZSPLIT"):
    from eyoc_amd import _lib
    _lib.knob("eyoc_maps_internal_order", 1)
    model.spconv_math = "split16"
for _ in range(3): pipe.register(batch)
pipe.timing = True; model.set_timing(True)
acc = {"feat": 0, "match": 0, "reg": 0}; fwd = 0; N = 10
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    pipe.register(batch)
    for k, v in pipe.stage_ms().items(): acc[k] += v / N
    fwd += sum(model.layer_ms()) / N
torch.cuda.synchronize()
print(f"{P} pairs: step {(time.perf_counter()-t0)/N*1e3:.2f} ms  stages {acc}  forward layers {fwd:.2f} ms  math {model.last_spconv_math}")
