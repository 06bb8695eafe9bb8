"""Phase timeline of the transposed kernel's workgroups on the 128 -> 64 layer (needs the -DEYOC_UP_TRACE build:
make -C eyoc_amd/csrc ../lib/libeyoc_hip_uptrace.so; EYOC_HIP_LIB=.../libeyoc_hip_uptrace.so python scripts/trace_up.py)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eyoc_amd, bench
from eyoc_amd import _lib
from eyoc_amd.harness import DeviceBatch, RegistrationConfig, RegistrationPipeline
dev = torch.device("cuda:0")
P = int(os.environ.get("PAIRS", "16"))
pairs = bench.make_pairs(list(range(P)))
model, sd = bench.build_model(dev, 0)
pipe = RegistrationPipeline(model, RegistrationConfig())
b = DeviceBatch(pairs, list(range(P)), dev, 5000, descriptor=dict(inlier_ratio=0.3))
raw = C.CDLL(_lib.LIB_PATH)
raw.eyoc_debug_up_trace.argtypes = [C.c_void_p, C.c_size_t]
NT = 8
for _ in range(2): pipe.features(b)
torch.cuda.synchronize()
buf = np.zeros(16384 * NT, np.uint64)
raw.eyoc_debug_up_trace(buf.ctypes.data, buf.size)        # clears
pipe.features(b); torch.cuda.synchronize()
raw.eyoc_debug_up_trace(buf.ctypes.data, buf.size)
t = buf.reshape(-1, NT).astype(np.int64)
t = t[t[:, 0] > 0]
xcc = t[:, 6] >> 32
spans = [t[xcc == x][:, 5].max() - t[xcc == x][:, 0].min() for x in np.unique(xcc)]
print(f"{len(t)} workgroups traced; XCD-local kernel spans {min(spans)}..{max(spans)} ticks (s_memtime)")
tick_us = 1 / 2100.0          # s_memtime counts shader clocks (~2.1 GHz under this load)
names = [("start -> header (n_unique, masks) read", 7, 0), ("header read -> stage issued (row list, DMA issue)", 1, 7), ("stage wait + barrier", 2, 1), ("offset loops (all blocks)", 3, 2),
         ("epilogue: rows + values", 4, 3), ("epilogue: stores issued", 5, 4), ("whole workgroup", 5, 0)]
for nm, a, c in names:
    v = (t[:, a] - t[:, c]) * tick_us
    print(f"  {nm:42s} mean {v.mean():7.2f} us  p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f}")
