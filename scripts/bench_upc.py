"""Transposed (3^3, stride 2) layers of the bench geometry: spconv_up.hip (Morton tiles) vs spconv_upc.hip (class-major tiles).
Checks both against an fp64 product of the decoded operands."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import eyoc_amd, bench
from eyoc_amd import _lib, synthetic as syn
from test_gpu_split16 import morton_order
pairs = int(os.environ.get("PAIRS", "64"))
ps = bench.make_pairs(list(range(pairs)))
clouds = []
for p in ps: clouds += [p["coords0"], p["coords1"]]
coords = syn.batch_coords(clouds)
coords = coords[morton_order(coords)]
lib = _lib.load()
for spec in os.environ.get("TILE_ROWS", "").split(","):
    if spec: _lib.check(lib.eyoc_spconv_upc_tile_rows(int(spec.split(":")[0]), int(spec.split(":")[1])))
def timeit(fn, reps=int(os.environ.get("REPS", "10"))):
    for _ in range(1 if os.environ.get("ONLY_UPC") else 3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
def layer_ref(T, x, W, rows):
    """fp64 reference of output rows `rows`: sum_k x[T[k, rows]] @ W[k]"""
    out = torch.zeros(len(rows), W.shape[2], dtype=torch.float64, device="cuda")
    for k in range(27):
        idx = T[k, rows].long(); m = idx >= 0
        if bool(m.any()): out[m] += x[idx[m]].double() @ W[k].double()
    return out
res = {}
for mode in ((2,) if os.environ.get("ONLY_UPC") else (1, 2)):
    _lib.knob("eyoc_spconv_select_up_kernel", 1)        # maps with spconv_up.hip's records (mode 2 builds its own workspace below)
    cm = eyoc_amd.CoordinateManager(torch.from_numpy(coords).cuda())
    maps = cm.maps(); info = cm.info()
    for lvl, cin, cout in ((0, 128, 64), (1, 256, 64), (2, 256, 128)):
        n = info["rows"][lvl]; n_in = info["rows"][lvl + 1]
        tab = lib.eyoc_maps_table(maps, 2, lvl)
        g = torch.Generator(device="cuda").manual_seed(lvl)
        x = torch.randn(n_in, cin, device="cuda", generator=g); xs = torch.empty_like(x)
        lib.eyoc_split16_encode(_lib.ctx(), _lib.ptr(x), n_in, cin, cin, _lib.ptr(xs), cin, _lib.stream_ptr())
        W = np.random.default_rng(lvl).normal(size=(27, cin, cout)).astype(np.float32)
        packed = np.zeros(W.size, np.float32); osc = np.ones(1, np.float32)
        lib.eyoc_spconv_pack_weights_split16(W.ctypes.data, None, 27, cin, cout, packed.ctypes.data, osc.ctypes.data)
        wd = torch.from_numpy(packed).cuda(); osd = torch.from_numpy(osc).cuda()
        out = torch.full((n, cout), float("nan"), device="cuda")
        if mode == 1:
            # through the model's path: eyoc_spconv_ex on the table picks spconv_up.hip only via the maps; time the gather kernel here and
            # take spconv_up.hip's time from profiles/r4_layer_times.txt (0.64 / 0.94 / 1.26 ms)
            run = lambda: _lib.check(lib.eyoc_spconv_ex(_lib.ctx(), tab, 27, n, n_in, _lib.ptr(xs), cin, cin, _lib.ptr(wd), cout, None, None, 0, 0, _lib.ptr(out), cout, 1, 0, _lib.ptr(osd), _lib.stream_ptr()))
            t_build = 0.0
        else:
            ws = torch.zeros(int(lib.eyoc_spconv_upc_bytes(n)) + 256, dtype=torch.uint8, device="cuda")
            wsp = (_lib.ptr(ws) + 255) & ~255 if isinstance(_lib.ptr(ws), int) else None
            import ctypes as C
            base = ws.data_ptr(); al = (base + 255) & ~255
            hinfo = np.zeros(19, np.int32)
            _lib.check(lib.eyoc_spconv_upc_build(_lib.ctx(), tab, n, C.c_void_p(al), hinfo.ctypes.data, _lib.stream_ptr()))
            t_build = timeit(lambda: _lib.check(lib.eyoc_spconv_upc_build(_lib.ctx(), tab, n, C.c_void_p(al), None, _lib.stream_ptr())))
            print(f"  lvl{lvl}: tiles {hinfo[0]} (at most {(n + 127) // 128} + 8), class rows {hinfo[10:18].tolist()}, overflowed tiles {hinfo[18]}", flush=True)
            run = lambda: _lib.check(lib.eyoc_spconv_upc(_lib.ctx(), tab, C.c_void_p(al), n, n_in, _lib.ptr(xs), cin, cin, _lib.ptr(wd), cout, None, 0, _lib.ptr(out), cout, 0, _lib.ptr(osd), _lib.stream_ptr()))
        t = timeit(run)
        torch.cuda.synchronize()
        T = torch.empty(27 * n, dtype=torch.int32, device="cuda")
        _lib.check(lib.eyoc_maps_copy_table(maps, 2, lvl, _lib.ptr(T), _lib.stream_ptr())); T = T.view(27, n)
        rows = torch.randint(0, n, (20000,), device="cuda")
        if os.environ.get("ONLY_UPC"):
            err = float("nan")
        else:
            ref = layer_ref(T, x, torch.from_numpy(W).cuda(), rows)
            err = float((out[rows].double() - ref).abs().max() / ref.abs().max())
        nan = int(torch.isnan(out).any(dim=1).sum())
        res[(mode, lvl)] = (t, t_build, err, nan)
        print(f"mode {mode} lvl{lvl} {cin}->{cout} n={n}: {t:.3f} ms (build {t_build:.3f} ms) rel err vs fp64 {err:.2e} rows never written {nan}", flush=True)
