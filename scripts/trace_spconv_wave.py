"""Phase timeline of a few waves of the wave-private spconv kernel (needs a -DEYOC_TRACE build: EYOC_HIP_LIB)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eyoc_amd  # noqa: E402
from eyoc_amd import _lib, synthetic as syn  # noqa: E402

clouds = []
for s in range(int(os.environ.get("PAIRS", "8"))):
    p = syn.make_pair(s)
    clouds += [p["coords0"], p["coords1"]]
cm = eyoc_amd.CoordinateManager(torch.from_numpy(syn.batch_coords(clouds)).cuda())
maps = cm.maps()
lib = _lib.load()
info = cm.info()
lvl, cin, cout = int(os.environ.get("LVL", "1")), 64, 64
n = info["rows"][lvl]
tab = lib.eyoc_maps_table(maps, 0, lvl)
x = torch.randn(n, cin, device="cuda")
W = np.random.default_rng(0).normal(size=(27, cin, cout)).astype(np.float32)
packed = np.zeros(W.size, np.float32)
lib.eyoc_spconv_pack_weights(W.ctypes.data, None, 27, cin, cout, packed.ctypes.data)
wd = torch.from_numpy(packed).cuda()
out = torch.empty(n, cout, device="cuda")
for _ in range(3):
    _lib.check(lib.eyoc_spconv(_lib.ctx(), tab, 27, n, _lib.ptr(x), cin, cin, _lib.ptr(wd), cout, None, None, 0, 0, _lib.ptr(out), cout, _lib.stream_ptr()))
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
NW, NS = 16, 512
buf = np.zeros(NW * NS, np.uint64)
raw.eyoc_debug_trace_wave.argtypes = [C.c_void_p, C.c_size_t]
assert raw.eyoc_debug_trace_wave(buf.ctypes.data, buf.size) == 0
t = buf.reshape(NW, NS).astype(np.int64)
for w in range(0, NW, 3):
    st = t[w]
    k = int((st > 0).sum())
    st = st[:k]
    d = np.diff(st)
    body = d[2:-2]
    per = body[: (len(body) // 4) * 4].reshape(-1, 4)   # [next-unit logic + load issue, wait + MFMA, flush, (loop)]
    print(f"wave {w}: {k} stamps total {st[-1] - st[0]} ticks; prologue {d[0]} {d[1]}; units {len(per)}; epilogue {d[-1]}")
    print("  mean ticks per phase [prep, mfma, flush, loop]:", per.mean(0).round(0), "sum", per.mean(0).sum().round(0))
    big = per[per[:, 1] > np.median(per[:, 1]) * 1.3]
    print("  first 8 units:", per[:8].tolist())
    print("  mfma-phase histogram:", np.percentile(per[:, 1], [5, 25, 50, 75, 95]).round(0), " prep:", np.percentile(per[:, 0], [5, 50, 95]).round(0), " flush:", np.percentile(per[:, 2], [5, 50, 95]).round(0))
