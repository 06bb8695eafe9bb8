"""SC2-PCR back-end alone at the reference's KITTI setting (8000 resampled correspondences per pair)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import eyoc_amd  # noqa: E402
import _inputs as gi  # noqa: E402

m = eyoc_amd.Matcher(inlier_threshold=0.6, d_thre=0.1, ratio=0.2, nms_radius=0.6, max_points=8000, k1=30, k2=20, num_iterations=20)
T = gi.rigid(0.02, -0.01, 0.1, 4.0, 0.3, -0.2)
B = int(os.environ.get("PAIRS", "8"))
src, tgt = [], []
for b in range(B):
    p0, p1, _ = gi.corr_case(500 + b, 8000, T, 0.2, noise=0.05)
    src.append(torch.from_numpy(p0).cuda()); tgt.append(torch.from_numpy(p1).cuda())
for _ in range(2):
    m.SC2_PCR_batch(src, tgt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    m.SC2_PCR_batch(src, tgt)
torch.cuda.synchronize()
print(f"SC2-PCR batched: {(time.perf_counter() - t0) / 3 / B * 1e3:.3f} ms per pair ({B} pairs)")
t0 = time.perf_counter()
for _ in range(3):
    m.SC2_PCR(src[0][None], tgt[0][None])
torch.cuda.synchronize()
print(f"SC2-PCR single:  {(time.perf_counter() - t0) / 3 * 1e3:.3f} ms per pair")
