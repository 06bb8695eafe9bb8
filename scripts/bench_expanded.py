"""ResUNetExpBN2C (model/resunet.py:487-490) eval forward on a batch of PAIRS synthetic pairs: the packed plan (eyoc_model_desc.expanded,
what ``model(x)`` runs) next to the layer-by-layer path (eyoc_amd/train.py forward_layers under no_grad - what eval mode ran up to
round 5) and the ResUNetBN2C forward on the same batch.  usage: PAIRS=16 python scripts/bench_expanded.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, eyoc_amd
from eyoc_amd import synthetic as syn
from eyoc_amd.train import forward_layers

P = int(os.environ.get("PAIRS", "16"))
dev = torch.device("cuda:0")
pairs = bench.make_pairs(list(range(P)))
coords = syn.batch_coords([c for p in pairs for c in (p["coords0"], p["coords1"])])
x = eyoc_amd.SparseTensor(torch.ones((len(coords), 1), device=dev), coordinates=torch.from_numpy(coords).to(dev))


def model_of(name, sd):
    m = eyoc_amd.load_model(name)(1, 32, bn_momentum=0.05, conv1_kernel_size=5, normalize_feature=True)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return m.to(dev).eval()


def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


exp, plain = model_of("ResUNetExpBN2C", syn.make_weights(seed=33, expanded=True)), model_of("ResUNetBN2C", syn.make_weights())
with torch.no_grad():
    a, b = exp(x).F, forward_layers(exp, x).F
    print(f"{P} pairs, {len(coords)} voxels: packed plan vs layer by layer: max |diff| {float((a - b).abs().max()):.2e}")
    t_plain, t_packed, t_layers = timed(lambda: plain(x)), timed(lambda: exp(x)), timed(lambda: forward_layers(exp, x), 3)
exp.set_timing(True); exp(x)
ms = exp.layer_ms()
names = [w["name"] for w in exp.layer_work(x)]
print(f"ResUNetBN2C {t_plain:.2f} ms | ResUNetExpBN2C packed plan {t_packed:.2f} ms ({exp.last_spconv_math}), layer by layer {t_layers:.2f} ms")
print("stand-alone norms: " + ", ".join(f"{n} {t * 1e3:.0f} us" for n, t in zip(names, ms) if n.startswith("norm")))
