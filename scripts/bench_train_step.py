"""Wall time of one training iteration (maps + train-mode forward + hardest-contrastive loss + backward + SGD step) on two
synthetic clouds; ``BEAMS=64 AZ=2000`` = two 30k-voxel clouds (the bench's `train_step_ms`), the default two 11k-voxel ones."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eyoc_amd
from eyoc_amd import synthetic as syn
from eyoc_amd.autograd import contrastive_hardest_negative_loss

BEAMS, AZ = int(os.environ.get("BEAMS", "32")), int(os.environ.get("AZ", "1000"))
ITERS = int(os.environ.get("ITERS", "30"))
dev = torch.device("cuda:0")
if os.environ.get("SPCONV_KERNEL"):      # eyoc_spconv_select_kernel: 0 workgroup-tiled, 1 wave-private (default: by size)
    from eyoc_amd import _lib
    _lib.knob("eyoc_spconv_select_kernel", int(os.environ["SPCONV_KERNEL"]))
p = syn.make_pair(1000, beams=BEAMS, azimuths=AZ, band=None)
from scipy.spatial import cKDTree
T = np.asarray(p["T_gt"], np.float64)
d, j = cKDTree(p["xyz1"].astype(np.float64)).query(p["xyz0"].astype(np.float64) @ T[:3, :3].T + T[:3, 3])
i = np.nonzero(d < 0.3)[0]
pos = torch.from_numpy(np.stack([i, j[i]], 1))
model = eyoc_amd.load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, conv1_kernel_size=5, normalize_feature=True).to(dev).train()
opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.8, weight_decay=1e-4)
coords = torch.from_numpy(syn.batch_coords([p["coords0"], p["coords1"]])).to(dev)
feats = torch.ones((coords.shape[0], 1), device=dev)
n0 = len(p["coords0"])

def step(it):
    out = model(eyoc_amd.SparseTensor(feats, coordinates=coords)).F
    np.random.seed(it)
    lp, ln = contrastive_hardest_negative_loss(out[:n0], out[n0:], pos, num_pos=1024, num_hn_samples=2048)
    loss = lp + ln
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss

for it in range(5): step(it)
torch.cuda.synchronize(); t0 = time.perf_counter()
tf = tl = tb = 0.0
for it in range(ITERS):
    a = time.perf_counter()
    out = model(eyoc_amd.SparseTensor(feats, coordinates=coords)).F
    b = time.perf_counter()
    np.random.seed(it)
    lp, ln = contrastive_hardest_negative_loss(out[:n0], out[n0:], pos, num_pos=1024, num_hn_samples=2048)
    loss = lp + ln
    c = time.perf_counter()
    opt.zero_grad(); loss.backward(); opt.step()
    e = time.perf_counter()
    tf += b - a; tl += c - b; tb += e - c
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / ITERS * 1e3
print(f"{coords.shape[0]} voxels: {wall:.2f} ms per iteration (host enqueue: forward {tf / ITERS * 1e3:.2f}, loss {tl / ITERS * 1e3:.2f}, backward + step {tb / ITERS * 1e3:.2f} ms)")
