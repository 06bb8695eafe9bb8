import os, sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import eyoc_amd
from eyoc_amd import _lib, synthetic as syn
from eyoc_amd.harness import DeviceBatch, RegistrationConfig, RegistrationPipeline

dev = torch.device("cuda:0")
model = eyoc_amd.load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, conv1_kernel_size=5, normalize_feature=True)
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_weights().items()})
model = model.to(dev).eval()
cfg = RegistrationConfig()
pipe = RegistrationPipeline(model, cfg)
pairs = [syn.make_pair(s) for s in range(32)]
full = DeviceBatch(pairs, list(range(32)), dev, cfg.n_points)
h1 = DeviceBatch(pairs[:16], list(range(16)), dev, cfg.n_points)
h2 = DeviceBatch(pairs[16:], list(range(16, 32)), dev, cfg.n_points)

# a context per stream: scratch buffers must not be shared by concurrent streams
_orig_ctx = _lib.ctx
_ctxs = {}
def ctx_per_stream(device_index=None):
    key = torch.cuda.current_stream().cuda_stream
    if key not in _ctxs:
        h = C.c_void_p()
        _lib.check(_lib.load().eyoc_create(0, C.byref(h)), "eyoc_create")
        _ctxs[key] = h
    return _ctxs[key]

def run_single(n):
    for _ in range(2): pipe.register(full, return_device=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): pipe.register(full, return_device=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

def run_dual(n, b1=None, b2=None):
    b1 = b1 or h1; b2 = b2 or h2
    _lib.ctx = ctx_per_stream
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def step():
        with torch.cuda.stream(s1): pipe.register(b1, return_device=True)
        with torch.cuda.stream(s2): pipe.register(b2, return_device=True)
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n * 1e3
    _lib.ctx = _orig_ctx
    return t

print("single stream, 32 pairs per step: %.2f ms" % run_single(8))
print("two streams, 16 + 16 pairs per step: %.2f ms" % run_dual(8))
print("single again: %.2f ms" % run_single(8))

full2 = DeviceBatch(pairs, list(range(32)), dev, cfg.n_points)
t = run_dual(6, full, full2)
print("two streams, 32 + 32 pairs per double step: %.2f ms = %.2f ms per 32 pairs" % (t, t / 2))
