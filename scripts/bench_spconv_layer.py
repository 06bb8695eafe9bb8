"""Micro-benchmark of single sparse-conv layers on the bench geometry (diagnostics)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eyoc_amd  # noqa: E402
from eyoc_amd import _lib, synthetic as syn  # noqa: E402

pairs = int(os.environ.get("PAIRS", "8"))
clouds = []
for s in range(pairs):
    p = syn.make_pair(s)
    clouds += [p["coords0"], p["coords1"]]
coords = torch.from_numpy(syn.batch_coords(clouds)).cuda()
cm = eyoc_amd.CoordinateManager(coords)
maps = cm.maps()
lib = _lib.load()
# kernel selection through the library's setters (it reads no environment variables): WAVE=0/1 forces the workgroup-tiled /
# wave-private kernel, RS=0/2 the wave-private / row-stationary split16 kernel
if os.environ.get("WAVE"):
    _lib.knob("eyoc_spconv_select_kernel", int(os.environ["WAVE"]))
if os.environ.get("RS"):
    _lib.knob("eyoc_spconv_select_split16_kernel", int(os.environ["RS"]))
info = cm.info()
print("rows", info["rows"], "pairs_s1", info["pairs_s1"], flush=True)
MATH = int(os.environ.get("MATH", "1"))
cfgs = [("s1", 1, 64, 64), ("s1", 0, 64, 64), ("s1", 2, 128, 128), ("up", 0, 128, 64), ("s1", 0, 32, 32), ("s1", 3, 256, 256), ("s1", 2, 64, 64)]
if os.environ.get("ONE"):
    cfgs = cfgs[:1]
if os.environ.get("K8"):
    cfgs = [("up", 0, 128, 64), ("down", 0, 32, 64), ("up", 1, 256, 64), ("down", 1, 64, 128)]
for kind, lvl, cin, cout in cfgs:
    n_out = info["rows"][lvl]
    if kind == "down":
        n_out = info["rows"][lvl + 1]
    n_in = info["rows"][lvl + 1] if kind == "up" else info["rows"][lvl]
    tab = lib.eyoc_maps_table(maps, {"s1": 0, "down": 1, "up": 2}[kind], lvl)
    prs = info["pairs_s1"][lvl] if kind == "s1" else info["pairs_up"][lvl] if kind == "up" else info["pairs_down"][lvl]
    x = torch.randn(n_in, cin, device="cuda")
    W = np.random.default_rng(0).normal(size=(27, cin, cout)).astype(np.float32)
    packed = np.zeros(W.size, np.float32)
    osc = np.ones(1, np.float32)
    if MATH:
        lib.eyoc_spconv_pack_weights_split16(W.ctypes.data, None, 27, cin, cout, packed.ctypes.data, osc.ctypes.data)
        xs = torch.empty_like(x)
        lib.eyoc_split16_encode(_lib.ctx(), _lib.ptr(x), n_in, cin, cin, _lib.ptr(xs), cin, _lib.stream_ptr())
        x = xs
    else:
        lib.eyoc_spconv_pack_weights(W.ctypes.data, None, 27, cin, cout, packed.ctypes.data)
    wd = torch.from_numpy(packed).cuda()
    osd = torch.from_numpy(osc).cuda()
    out = torch.empty(n_out, cout, device="cuda")
    # the tiling order the model forward would use for this table
    perm = torch.empty(n_out, dtype=torch.int32, device="cuda")

    def run():
        _lib.check(lib.eyoc_spconv_ex(_lib.ctx(), tab, 27, n_out, n_in, _lib.ptr(x), cin, cin, _lib.ptr(wd), cout, None, None, 0, 0,
                                      _lib.ptr(out), cout, MATH, MATH, _lib.ptr(osd), _lib.stream_ptr()))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{kind} lvl{lvl} {cin}->{cout} n_out={n_out} pairs={prs}: {ms:.3f} ms  {2 * prs * cin * cout / ms / 1e9:.1f} TFLOP/s", flush=True)
