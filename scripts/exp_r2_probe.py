"""Round-2 probes on the GPU: (a) stage times of the registration step with planted descriptors at several inlier
ratios, (b) does a Morton (Z-order) row order of the INPUT cloud speed the sparse convolutions up."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eyoc_amd
from eyoc_amd import synthetic as syn
from eyoc_amd.harness import DeviceBatch, RegistrationConfig, RegistrationPipeline
from eyoc_amd.eval import gather_rows, knn1_segmented
from eyoc_amd import registration as reg

P = int(os.environ.get("P", "16"))
dev = torch.device("cuda:0")
sd = syn.make_weights()
model = eyoc_amd.load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, conv1_kernel_size=5, normalize_feature=True)
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
model = model.to(dev).eval()
seeds = list(range(P))
t0 = time.time(); pairs = [syn.make_pair(s) for s in seeds]; print(f"{P} pairs generated in {time.time()-t0:.1f}s", flush=True)
pipe = RegistrationPipeline(model, RegistrationConfig())

def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e

RATIOS = [None if r == "none" else float(r) for r in os.environ.get("RATIOS", "none,0.05,0.15,0.3,0.6").split(",")]
for ratio in RATIOS:
    batch = DeviceBatch(pairs, seeds, dev, 5000, descriptor=None if ratio is None else dict(inlier_ratio=ratio))
    for _ in range(2): res = pipe.register(batch)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0 = ev(); F = pipe.features(batch).F; e1 = ev()
        F0 = gather_rows(F, batch.sel0, batch.G0, batch.beta); F1 = gather_rows(F, batch.sel1, batch.G1, batch.beta); e2 = ev()
        nn = knn1_segmented(F0, F1, batch.seg, batch.seg, "SquareL2", return_distance=False); e3 = ev()
        r = reg.ransac_batched_from_correspondences(batch.xyz0.reshape(-1, 3), batch.xyz1.reshape(-1, 3), nn, batch.seg, batch.seg, 0.3, 4000000, seed=0); e4 = ev()
        torch.cuda.synchronize()
        ts.append([e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3), e3.elapsed_time(e4)])
    ts = np.median(np.array(ts), 0)
    pipe.last_nn_idx = nn
    ir = pipe.correspondence_inlier_ratio(batch)
    evals = pipe.evaluate(batch, res)
    print(f"ratio={ratio}: forward {ts[0]:.2f} gather {ts[1]:.3f} nn {ts[2]:.2f} ransac {ts[3]:.2f} ms | realised inlier ratio {np.mean(ir):.3f} "
          f"survivors/pair {np.mean([x.survivors for x in res]):.0f} inliers {np.mean([x.inliers for x in res]):.0f} "
          f"success {np.mean([e['success'] for e in evals]):.3f} rte {np.median([e['rte'] for e in evals]):.3f} rre {np.median([e['rre_deg'] for e in evals]):.3f}", flush=True)

if os.environ.get("NO_MORTON"):
    sys.exit(0)
# (b) Morton order of the input rows
def morton_perm(coords):
    c = coords.astype(np.int64)
    b = c[:, 0]; x = c[:, 1] + (1 << 17); y = c[:, 2] + (1 << 17); z = c[:, 3] + (1 << 17)
    def spread(v):
        out = np.zeros_like(v)
        for i in range(18): out |= ((v >> i) & 1) << (3 * i)
        return out
    key = (b << 54) | spread(x) | (spread(y) << 1) | (spread(z) << 2)
    return np.argsort(key, kind="stable")

batch = DeviceBatch(pairs, seeds, dev, 5000)
coords = batch.coords.cpu().numpy(); feats = batch.feats
for name, perm in (("sweep order", None), ("morton order", morton_perm(coords)), ("random order", np.random.default_rng(0).permutation(len(coords)))):
    if perm is not None:
        # keep each cloud contiguous? morton key has the batch in the top bits, random does not (worst case)
        c = torch.from_numpy(coords[perm]).to(dev)
    else:
        c = batch.coords
    x = eyoc_amd.SparseTensor(feats, coordinates=c)
    model.set_timing(True)
    for _ in range(2): model(x)
    ms = np.zeros(model._handle and 23 or 23)
    acc = None
    for _ in range(3):
        x = eyoc_amd.SparseTensor(feats, coordinates=c)
        torch.cuda.synchronize(); t0 = time.perf_counter(); model(x); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
        m = np.array(model.layer_ms()); acc = m if acc is None else acc + m
    acc /= 3
    model.set_timing(False)
    work = model.layer_work(x)
    print(f"{name}: forward kernels {acc.sum():.2f} ms (wall incl. maps {wall:.2f}); " + " ".join(f"{w['name']}={a:.2f}" for w, a in zip(work, acc)), flush=True)
