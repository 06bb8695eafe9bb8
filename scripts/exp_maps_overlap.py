"""Experiment: the map build of the NEXT batch on a side stream while the RANSAC of the current batch runs (latency-bound integer
kernels next to fp64-VALU-bound ones)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eyoc_amd, bench
from eyoc_amd import registration as reg, synthetic as syn
from eyoc_amd.eval import gather_rows, knn1_segmented
from eyoc_amd.harness import DeviceBatch, RegistrationConfig, RegistrationPipeline
dev = torch.device("cuda:0")
P = 64
pairs = bench.make_pairs(list(range(P)))
model, sd = bench.build_model(dev, 0)
cfg = RegistrationConfig()
pipe = RegistrationPipeline(model, cfg)
b = DeviceBatch(pairs, list(range(P)), dev, 5000, descriptor=dict(inlier_ratio=0.3))
F = pipe.features(b).F
F0 = gather_rows(F, b.sel0, b.G0, b.beta); F1 = gather_rows(F, b.sel1, b.G1, b.beta)
nn = knn1_segmented(F0, F1, b.seg, b.seg, "SquareL2", return_distance=False)
clouds = []
for p in pairs: clouds += [p["coords0"], p["coords1"]]
coords = torch.from_numpy(syn.batch_coords(clouds)).to(dev)
def ransac(): return reg.ransac_batched_from_correspondences(b.xyz0.reshape(-1, 3), b.xyz1.reshape(-1, 3), nn, b.seg, b.seg, 0.3, 4000000, seed=0)
def maps(): return eyoc_amd.CoordinateManager(coords).maps()
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("ransac alone", round(t(ransac), 2), "ms; maps alone", round(t(maps), 2), "ms")
side = torch.cuda.Stream()
def both():
    r = ransac()                       # enqueued on the main stream (asynchronous)
    with torch.cuda.stream(side):
        m = maps()                     # the build synchronises its own stream twice
    torch.cuda.current_stream().wait_stream(side)
    return r, m
print("ransac (main stream) + maps (side stream)", round(t(both), 2), "ms")
def seq():
    r = ransac(); m = maps(); return r, m
print("sequential", round(t(seq), 2), "ms")
