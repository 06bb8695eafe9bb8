"""Phase timeline of a few spconv workgroups (needs the -DEYOC_TRACE build copied over libeyoc_hip.so)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eyoc_amd  # noqa: E402
from eyoc_amd import _lib, synthetic as syn  # noqa: E402

clouds = []
for s in range(8):
    p = syn.make_pair(s)
    clouds += [p["coords0"], p["coords1"]]
cm = eyoc_amd.CoordinateManager(torch.from_numpy(syn.batch_coords(clouds)).cuda())
maps = cm.maps()
lib = _lib.load()
info = cm.info()
lvl, cin, cout = 1, 64, 64
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
NB, NW, NS = 8, 8, 512
buf = np.zeros(NB * NW * NS, np.uint64)
raw.eyoc_debug_trace.argtypes = [C.c_void_p, C.c_size_t]
assert raw.eyoc_debug_trace(buf.ctypes.data, buf.size) == 0
t = buf.reshape(NB, NW, NS).astype(np.int64)
for b in (0, 1):
    for w in (0, 3):
        st = t[b, w]
        k = int((st > 0).sum())
        st = st[:k]
        d = np.diff(st)
        print(f"block {300 + b} wave {w}: {k} stamps, total {st[-1] - st[0]} ticks")
        print("  start->phase0:", d[0], " phase0->loop:", d[1] if k > 2 else None)
        NP = int(os.environ.get("NPH", "3"))
        body = d[2:-2]
        per = body[: (len(body) // NP) * NP].reshape(-1, NP)     # phases between the TR() stamps of one iteration
        print("  items:", len(per), "mean ticks per phase:", per.mean(0).round(0), "sum", per.mean(0).sum().round(0))
        print("  first 6 items:", per[:6].tolist())
        print("  tail:", d[-2:])
