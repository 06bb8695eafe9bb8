"""Strided 3^3 / stride-2 split16 layers on the bench geometry: gathering kernel vs the 64-row-tile staged kernel (diagnostics)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import eyoc_amd, bench
from eyoc_amd import _lib, synthetic as syn
from test_gpu_split16 import morton_order
pairs = int(os.environ.get("PAIRS", "16"))
ps = bench.make_pairs(list(range(pairs)))
clouds = []
for p in ps: clouds += [p["coords0"], p["coords1"]]
coords = syn.batch_coords(clouds)
coords = coords[morton_order(coords)]
cm = eyoc_amd.CoordinateManager(torch.from_numpy(coords).cuda())
maps = cm.maps(); lib = _lib.load(); info = cm.info()
print("rows", info["rows"], flush=True)
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
for lvl, cin, cout in ((0, 32, 64), (1, 64, 128), (2, 128, 256)):
    n_in, n = info["rows"][lvl], info["rows"][lvl + 1]; prs = info["pairs_down"][lvl]
    tab = lib.eyoc_maps_table(maps, 1, lvl)
    x = torch.randn(n_in, cin, device="cuda"); xs = torch.empty_like(x)
    lib.eyoc_split16_encode(_lib.ctx(), _lib.ptr(x), n_in, cin, cin, _lib.ptr(xs), cin, _lib.stream_ptr())
    W = np.random.default_rng(0).normal(size=(27, cin, cout)).astype(np.float32)
    packed = np.zeros(W.size, np.float32); osc = np.ones(1, np.float32)
    lib.eyoc_spconv_pack_weights_split16(W.ctypes.data, None, 27, cin, cout, packed.ctypes.data, osc.ctypes.data)
    wd = torch.from_numpy(packed).cuda(); osd = torch.from_numpy(osc).cuda()
    out = torch.empty(n, cout, device="cuda"); out2 = torch.empty(n, cout, device="cuda")
    local = torch.zeros(int(lib.eyoc_spconv_local_rulebook_bytes_tile(n, 64)), dtype=torch.uint8, device="cuda")
    ovf = torch.zeros(1, dtype=torch.int32, device="cuda")
    t_lr = timeit(lambda: lib.eyoc_spconv_build_local_rulebook_tile(_lib.ctx(), tab, 27, n, 64, _lib.ptr(local), _lib.ptr(ovf), _lib.stream_ptr()))
    res = {}
    for name, mode in (("wave", 0), ("auto", 1)):
        lib.eyoc_spconv_select_split16_kernel(mode)
        res[name] = timeit(lambda: _lib.check(lib.eyoc_spconv_ex(_lib.ctx(), tab, 27, n, n_in, _lib.ptr(xs), cin, cin, _lib.ptr(wd), cout, None, None, 0, 0, _lib.ptr(out), cout, 1, 0, _lib.ptr(osd), _lib.stream_ptr())))
    lib.eyoc_spconv_select_split16_kernel(1)
    res["st64"] = timeit(lambda: _lib.check(lib.eyoc_spconv_staged_tile(_lib.ctx(), tab, _lib.ptr(local), 64, n, n_in, _lib.ptr(xs), cin, cin, _lib.ptr(wd), cout, None, None, 0, 0, _lib.ptr(out2), cout, 0, _lib.ptr(osd), _lib.stream_ptr())))
    torch.cuda.synchronize()
    d = float((out - out2).abs().max() / out.abs().max())
    # fp64 check of 512 sampled output rows
    tabh = torch.empty((27, n), dtype=torch.int32, device="cuda")
    _lib.check(lib.eyoc_maps_copy_table(maps, 1, lvl, _lib.ptr(tabh), _lib.stream_ptr()))
    rows = torch.randint(0, n, (512,), device="cuda")
    nb = tabh[:, rows].long()                                     # [27, 512]
    xg = torch.where((nb >= 0)[..., None], x.double()[nb.clamp(min=0)], torch.zeros((), dtype=torch.float64, device="cuda"))
    ref = torch.einsum("kri,kio->ro", xg, torch.from_numpy(W).cuda().double())
    e1 = float((out[rows].double() - ref).abs().max() / ref.abs().max()); e2 = float((out2[rows].double() - ref).abs().max() / ref.abs().max())
    d = f"{d:.1e} (vs fp64 on 512 rows: gather {e1:.1e}, st64 {e2:.1e})"
    print(f"down lvl{lvl} {cin}->{cout} n_in={n_in} n_out={n} pairs={prs} overflow={int(ovf.item())} records {t_lr:.3f} ms | " + "  ".join(f"{k} {v:.3f} ms ({2*prs*cin*cout/v/1e9:.0f} TF)" for k, v in res.items()) + f" | st64 vs gather {d}", flush=True)
