"""End-to-end exercise of the training path: a few hundred SGD steps of the hardest-contrastive loss
(lib/trainer.py:935-991, optimiser settings of config.py: SGD lr 0.1, momentum 0.8, weight decay 1e-4, exponential decay) on
synthetic KITTI-shaped pairs with ground-truth positives, through ``model.train()(x)`` = eyoc_amd/train.py; then the eval-mode
registration pipeline (fused kernels, 4-point RANSAC) on HELD-OUT pairs with the trained weights and NO planted descriptors.
Reports loss, feature-match recall (fraction of sampled source voxels whose feature nearest neighbour is within 0.3 m under the
ground truth = the inlier ratio RANSAC sees) and registration success (RTE < 2 m, RRE < 5 deg).  Diagnostics, not the benchmark."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eyoc_amd  # noqa: E402
from eyoc_amd import synthetic as syn  # noqa: E402
from eyoc_amd.autograd import contrastive_hardest_negative_loss  # noqa: E402
from eyoc_amd.harness import DeviceBatch, RegistrationConfig, RegistrationPipeline  # noqa: E402

ITERS = int(os.environ.get("ITERS", "400"))
N_TRAIN = int(os.environ.get("N_TRAIN", "24"))
BEAMS, AZ = int(os.environ.get("BEAMS", "32")), int(os.environ.get("AZ", "1000"))
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)


def positives(p, radius=0.3):
    from scipy.spatial import cKDTree
    T = np.asarray(p["T_gt"], np.float64)
    d, j = cKDTree(p["xyz1"].astype(np.float64)).query(p["xyz0"].astype(np.float64) @ T[:3, :3].T + T[:3, 3])
    i = np.nonzero(d < radius)[0]
    return np.stack([i, j[i]], 1)


t0 = time.time()
train = [syn.make_pair(1000 + s, beams=BEAMS, azimuths=AZ, band=None) for s in range(N_TRAIN)]
held = [syn.make_pair(5000 + s, beams=BEAMS, azimuths=AZ, band=None) for s in range(8)]
pos = [positives(p) for p in train]
print(f"{N_TRAIN} training pairs ({np.mean([len(p['coords0']) for p in train]):.0f} voxels per cloud, {np.mean([len(q) for q in pos]):.0f} positives), "
      f"8 held-out, generated in {time.time() - t0:.1f} s", flush=True)

model = eyoc_amd.load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, conv1_kernel_size=5, normalize_feature=True).to(dev)
opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.8, weight_decay=1e-4)
sched = torch.optim.lr_scheduler.ExponentialLR(opt, 0.99)
cfg = RegistrationConfig(ransac_max_iteration=1000000, n_points=5000)


def evaluate(tag):
    model.eval()
    pipe = RegistrationPipeline(model, cfg)
    batch = DeviceBatch(held, list(range(8)), dev, cfg.n_points)          # no descriptor planting
    res = pipe.register(batch, seed=0)
    ratios = pipe.correspondence_inlier_ratio(batch)
    ev = pipe.evaluate(batch, res)
    print(f"[{tag}] held-out pairs: feature-match inlier ratio mean {np.mean(ratios):.3f} (min {np.min(ratios):.3f}), "
          f"registered {sum(e['success'] for e in ev)}/8, median RTE {np.median([e['rte'] for e in ev]):.3f} m, "
          f"median RRE {np.median([e['rre_deg'] for e in ev]):.3f} deg", flush=True)
    model.train()


evaluate("random init")
model.train()
t0 = time.time()
for it in range(ITERS):
    k = int(rng.integers(N_TRAIN))
    p, pp = train[k], pos[k]
    coords = torch.from_numpy(syn.batch_coords([p["coords0"], p["coords1"]])).to(dev)
    feats = torch.ones((coords.shape[0], 1), device=dev)
    out = model(eyoc_amd.SparseTensor(feats, coordinates=coords)).F
    n0 = len(p["coords0"])
    F0, F1 = out[:n0], out[n0:]
    np.random.seed(it)
    lp, ln = contrastive_hardest_negative_loss(F0, F1, torch.from_numpy(pp), num_pos=1024, num_hn_samples=2048)
    loss = lp + ln
    opt.zero_grad()
    loss.backward()
    opt.step()
    if (it + 1) % int(os.environ.get("LOG_EVERY", "50")) == 0:
        sched.step()
        torch.cuda.synchronize()
        print(f"iter {it + 1}: loss {float(loss.detach()):.4f} (pos {float(lp.detach()):.4f} neg {float(ln.detach()):.4f})  "
              f"{(time.time() - t0) / (it + 1) * 1e3:.0f} ms / iteration", flush=True)
    if (it + 1) % int(os.environ.get("EVAL_EVERY", "200")) == 0 or it + 1 == ITERS:
        evaluate(f"after {it + 1} iterations")
