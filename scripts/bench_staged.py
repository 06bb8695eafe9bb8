"""Stride-1 split16 layers on Morton-ordered rows: wave-private vs row-stationary vs staged kernel (diagnostics)."""
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
if os.environ.get("MORTON", "1") == "1": coords = coords[morton_order(coords)]
cm = eyoc_amd.CoordinateManager(torch.from_numpy(coords).cuda())
maps = cm.maps(); lib = _lib.load(); info = cm.info()
print("rows", info["rows"], flush=True)
LAYERS = tuple(tuple(int(v) for v in s.split(":")) for s in os.environ["LAYERS"].split(",")) if os.environ.get("LAYERS") else ((0, 64, 64), (2, 128, 128)) if os.environ.get("ONLY_ST") else ((0, 64, 64), (0, 32, 32), (1, 64, 64), (2, 128, 128), (3, 256, 256))
for lvl, cin, cout in LAYERS:
    n = info["rows"][lvl]; prs = info["pairs_s1"][lvl]
    tab = lib.eyoc_maps_table(maps, 0, lvl)
    x = torch.randn(n, cin, device="cuda") * float(os.environ.get("XSCALE", "1")); xs = torch.empty_like(x)   # XSCALE=0: all-zero activations (same instruction stream, fewer toggling bits: a power / clock probe)
    lib.eyoc_split16_encode(_lib.ctx(), _lib.ptr(x), n, cin, cin, _lib.ptr(xs), cin, _lib.stream_ptr())
    W = np.random.default_rng(0).normal(size=(27, cin, cout)).astype(np.float32)
    packed = np.zeros(W.size, np.float32); osc = np.ones(1, np.float32)
    lib.eyoc_spconv_pack_weights_split16(W.ctypes.data, None, 27, cin, cout, packed.ctypes.data, osc.ctypes.data)
    wd = torch.from_numpy(packed).cuda(); osd = torch.from_numpy(osc).cuda()
    out = torch.empty(n, cout, device="cuda")
    local = torch.zeros(int(lib.eyoc_spconv_local_rulebook_bytes(n)), dtype=torch.uint8, device="cuda")
    ovf = torch.zeros(1, dtype=torch.int32, device="cuda")
    def timeit(fn, reps=10):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
    t_lr = timeit(lambda: lib.eyoc_spconv_build_local_rulebook(_lib.ctx(), tab, 27, n, _lib.ptr(local), _lib.ptr(ovf), _lib.stream_ptr()))
    res_t = {}
    for name, mode in (() if os.environ.get("ONLY_ST") else (("wave", 0), ("rs", 2))):
        _lib.knob("eyoc_spconv_select_split16_kernel", mode)
        res_t[name] = timeit(lambda: _lib.check(lib.eyoc_spconv_ex(_lib.ctx(), tab, 27, n, n, _lib.ptr(xs), cin, cin, _lib.ptr(wd), cout, None, None, 0, 0, _lib.ptr(out), cout, 1, 1, _lib.ptr(osd), _lib.stream_ptr())))
    _lib.knob("eyoc_spconv_select_split16_kernel", 1)
    ref_out = None
    variants = [int(v) for v in os.environ.get("ST_VARIANTS", "1,100").split(",")]       # 100: variant 1 on records WITHOUT row grouping
    _lib.knob("eyoc_spconv_st_group_rows", 0)
    local_plain = torch.zeros_like(local)
    t_lr_plain = timeit(lambda: lib.eyoc_spconv_build_local_rulebook(_lib.ctx(), tab, 27, n, _lib.ptr(local_plain), _lib.ptr(ovf), _lib.stream_ptr()))
    _lib.knob("eyoc_spconv_st_group_rows", 1)
    run_plain = lambda: _lib.check(lib.eyoc_spconv_staged(_lib.ctx(), tab, _lib.ptr(local_plain), n, n, _lib.ptr(xs), cin, cin, _lib.ptr(wd), cout, None, _lib.ptr(res), 0 if res is None else cout, 0, _lib.ptr(out), cout, 1, _lib.ptr(osd), _lib.stream_ptr()))
    res = None
    if os.environ.get("RES") and cin == cout:      # residual layers (the second convolution of a block): RES=1
        res = torch.empty_like(x)
        lib.eyoc_split16_encode(_lib.ctx(), _lib.ptr(torch.randn(n, cout, device="cuda")), n, cout, cout, _lib.ptr(res), cout, _lib.stream_ptr())
    run_st = lambda: _lib.check(lib.eyoc_spconv_staged(_lib.ctx(), tab, _lib.ptr(local), n, n, _lib.ptr(xs), cin, cin, _lib.ptr(wd), cout, None, _lib.ptr(res), 0 if res is None else cout, 0, _lib.ptr(out), cout, 1, _lib.ptr(osd), _lib.stream_ptr()))
    best = {v: [] for v in variants}
    for rnd in range(int(os.environ.get("ROUNDS", "5"))):      # variants interleaved over several rounds: clocks drift with load
        for variant in variants:
            if variant == 100:
                _lib.knob("eyoc_spconv_select_st_kernel", 1)
                best[variant].append(timeit(run_plain, reps=5))
                if rnd == 0 and ref_out is not None and not torch.equal(ref_out, out): best[variant].append(-1e6)
                continue
            _lib.knob("eyoc_spconv_select_st_kernel", variant)
            best[variant].append(timeit(run_st, reps=5))
            if rnd == 0:
                torch.cuda.synchronize()
                if ref_out is None: ref_out = out.clone()
                elif not torch.equal(ref_out, out): best[variant].append(-1e6)   # a mismatch shows as an absurd time
    for v in variants: res_t[f"st{v}"] = float(np.median(best[v])); res_t[f"st{v}min"] = min(best[v])
    _lib.knob("eyoc_spconv_select_st_kernel", 1)
    print(f"lvl{lvl} {cin}->{cout} n={n} pairs={prs} overflow={int(ovf.item())} local-rulebook {t_lr:.3f} (in row order {t_lr_plain:.3f}) | " + "  ".join(f"{k} {v:.3f} ms ({2*prs*cin*cout/v/1e9:.0f} TF)" for k, v in res_t.items()), flush=True)
