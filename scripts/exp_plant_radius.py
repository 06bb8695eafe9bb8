import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import eyoc_amd
from eyoc_amd import synthetic as syn
from eyoc_amd.harness import DeviceBatch, RegistrationConfig, RegistrationPipeline
import bench
dev = torch.device("cuda:0")
pairs = bench.make_pairs(list(range(16)))
model, sd = bench.build_model(dev, 0)
pipe = RegistrationPipeline(model, RegistrationConfig())
for pr in (0.2, 0.3):
    b = DeviceBatch(pairs, list(range(16)), dev, 5000, descriptor=dict(inlier_ratio=0.3, plant_radius=pr))
    pipe.timing = True
    for _ in range(3): res = pipe.register(b)
    st = pipe.stage_ms(); ev = pipe.evaluate(b, res)
    print(pr, "reg ms", round(st["reg"], 2), "survivors", np.mean([r.survivors for r in res]), "inliers", np.mean([r.inliers for r in res]),
          "success", np.mean([e["success"] for e in ev]), "rte", np.median([e["rte"] for e in ev]), "planted", np.mean(b.planted))
