"""Experiment: software-pipeline a 64-pair step as two 32-pair halves - forward of half B on the main stream while the
matching + RANSAC of half A runs on a side stream (memory-bound vs VALU-bound kernels)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eyoc_amd, bench
from eyoc_amd import registration as reg
from eyoc_amd.eval import gather_rows, knn1_segmented
from eyoc_amd.harness import DeviceBatch, RegistrationConfig, RegistrationPipeline
dev = torch.device("cuda:0")
P = 64
pairs = bench.make_pairs(list(range(P)))
model, sd = bench.build_model(dev, 0)
cfg = RegistrationConfig()
pipe = RegistrationPipeline(model, cfg)
desc = dict(inlier_ratio=0.3)
full = DeviceBatch(pairs, list(range(P)), dev, 5000, descriptor=desc)
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("sequential 64:", round(t(lambda: pipe.register(full)), 2), "ms")
side = torch.cuda.Stream()
def match_reg(b, F, seed):
    F0 = gather_rows(F, b.sel0, b.G0, b.beta); F1 = gather_rows(F, b.sel1, b.G1, b.beta)
    nn = knn1_segmented(F0, F1, b.seg, b.seg, "SquareL2", return_distance=False)
    return reg.ransac_batched_from_correspondences(b.xyz0.reshape(-1, 3), b.xyz1.reshape(-1, 3), nn, b.seg, b.seg, 0.3, 4000000, seed=seed)
for nparts in (2, 4):
    h = P // nparts
    parts = [DeviceBatch(pairs[i*h:(i+1)*h], list(range(i*h, (i+1)*h)), dev, 5000, descriptor=desc) for i in range(nparts)]
    def overlapped():
        cur = torch.cuda.current_stream(); outs = []
        for i, b in enumerate(parts):
            F = pipe.features(b).F
            ev = torch.cuda.Event(); ev.record(cur)
            with torch.cuda.stream(side):
                side.wait_event(ev); F.record_stream(side)
                outs.append(match_reg(b, F, i * h))
        cur.wait_stream(side)
        return torch.cat(outs).cpu()
    print(f"overlapped {nparts} x {h}:", round(t(overlapped), 2), "ms")
    def seq_parts():
        outs = [match_reg(b, pipe.features(b).F, i * h) for i, b in enumerate(parts)]
        return torch.cat(outs).cpu()
    print(f"sequential {nparts} x {h}:", round(t(seq_parts), 2), "ms")
