import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import eyoc_amd, bench
from eyoc_amd import _lib, synthetic as syn
from test_gpu_split16 import morton_order
ps = bench.make_pairs(list(range(16)))
clouds = []
for p in ps: clouds += [p["coords0"], p["coords1"]]
coords = syn.batch_coords(clouds); coords = coords[morton_order(coords)]
lib = _lib.load()
cm = eyoc_amd.CoordinateManager(torch.from_numpy(coords).cuda()); maps = cm.maps(); info = cm.info()
for lvl, cin, cout in ((0, 128, 64), (1, 256, 64), (2, 256, 128)):
    n = info["rows"][lvl]; n_in = info["rows"][lvl + 1]
    tab = lib.eyoc_maps_table(maps, 2, lvl)
    x = torch.randn(n_in, cin, device="cuda"); xs = torch.empty_like(x)
    lib.eyoc_split16_encode(_lib.ctx(), _lib.ptr(x), n_in, cin, cin, _lib.ptr(xs), cin, _lib.stream_ptr())
    W = np.random.default_rng(lvl).normal(size=(27, cin, cout)).astype(np.float32)
    packed = np.zeros(W.size, np.float32); osc = np.ones(1, np.float32)
    lib.eyoc_spconv_pack_weights_split16(W.ctypes.data, None, 27, cin, cout, packed.ctypes.data, osc.ctypes.data)
    wd = torch.from_numpy(packed).cuda(); osd = torch.from_numpy(osc).cuda()
    ws = torch.zeros(int(lib.eyoc_spconv_upc_bytes(n)) + 256, dtype=torch.uint8, device="cuda"); al = (ws.data_ptr() + 255) & ~255
    outs = []
    for rep in range(6):
        _lib.check(lib.eyoc_spconv_upc_build(_lib.ctx(), tab, n, C.c_void_p(al), None, _lib.stream_ptr()))
        out = torch.empty(n, cout, device="cuda")
        _lib.check(lib.eyoc_spconv_upc(_lib.ctx(), tab, C.c_void_p(al), n, n_in, _lib.ptr(xs), cin, cin, _lib.ptr(wd), cout, None, 0, _lib.ptr(out), cout, 0, _lib.ptr(osd), _lib.stream_ptr()))
        torch.cuda.synchronize(); outs.append(out)
    print(lvl, "rows differing between rebuilds:", [int((outs[0] != o).any(dim=1).sum()) for o in outs[1:]], flush=True)
