"""Forward of small inputs on the two paths (fp32 kernels on the caller's rows vs Z-order + split16 + staged) - diagnostics."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, eyoc_amd
from eyoc_amd import _lib, synthetic as syn
lib = _lib.load()
dev = torch.device("cuda:0")
model, _ = bench.build_model(dev, 0)
p = syn.make_pair(0)
for name, clouds in (("half cloud", [p["coords0"][:15000]]), ("1 cloud", [p["coords0"]]), ("2 clouds", [p["coords0"], p["coords1"]])):
    coords = torch.from_numpy(syn.batch_coords(clouds)).to(dev)
    feats = torch.ones((coords.shape[0], 1), device=dev)
    for mode in ("old", "new"):
        _lib.knob("eyoc_maps_internal_order", 1 if mode == "new" else 0)
        model.spconv_math = "split16" if mode == "new" else "fp32"
        def run():
            return model(eyoc_amd.SparseTensor(feats, coordinates=coords)).F
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): run()
        torch.cuda.synchronize()
        print(f"{name} ({coords.shape[0]} rows) {mode}: maps + forward {(time.perf_counter()-t0)/20*1e3:.3f} ms", flush=True)
