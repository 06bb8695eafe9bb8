"""Transposed (kernel 3, stride 2) layers through the stride-1 staged kernel: records built from the TRANSPOSED table (a fine
row's 27 offsets -> coarse row or -1; 1-8 of them exist), stage = the tile's distinct coarse rows.  Diagnostics."""
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
cm = eyoc_amd.CoordinateManager(torch.from_numpy(coords).cuda())
maps = cm.maps(); lib = _lib.load(); info = cm.info()
print("rows", info["rows"], "pairs_up", info["pairs_up"], flush=True)
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
for lvl, cin, cout in ((0, 128, 64), (1, 256, 64), (2, 256, 128)):
    n = info["rows"][lvl]; n_in = info["rows"][lvl + 1]; prs = info["pairs_up"][lvl]
    tab = lib.eyoc_maps_table(maps, 2, lvl)
    x = torch.randn(n_in, cin, device="cuda"); xs = torch.empty_like(x)
    lib.eyoc_split16_encode(_lib.ctx(), _lib.ptr(x), n_in, cin, cin, _lib.ptr(xs), cin, _lib.stream_ptr())
    W = np.random.default_rng(0).normal(size=(27, cin, cout)).astype(np.float32)
    packed = np.zeros(W.size, np.float32); osc = np.ones(1, np.float32)
    lib.eyoc_spconv_pack_weights_split16(W.ctypes.data, None, 27, cin, cout, packed.ctypes.data, osc.ctypes.data)
    wd = torch.from_numpy(packed).cuda(); osd = torch.from_numpy(osc).cuda()
    ref = torch.empty(n, cout, device="cuda"); out = torch.empty(n, cout, device="cuda")
    run_ref = lambda: _lib.check(lib.eyoc_spconv_ex(_lib.ctx(), tab, 27, n, n_in, _lib.ptr(xs), cin, cin, _lib.ptr(wd), cout, None, None, 0, 0, _lib.ptr(ref), cout, 1, 1, _lib.ptr(osd), _lib.stream_ptr()))
    t_ref = timeit(run_ref)
    local = torch.zeros(int(lib.eyoc_spconv_local_rulebook_bytes(n)), dtype=torch.uint8, device="cuda")
    ovf = torch.zeros(1, dtype=torch.int32, device="cuda")
    res = {}
    variants = [int(v) for v in os.environ.get("ST_VARIANTS", "1").split(",")]
    for grp in [0, 1] + [100 + v for v in variants if v != 1]:
        _lib.knob("eyoc_spconv_st_group_rows", 1 if grp >= 100 else grp)
        _lib.knob("eyoc_spconv_select_st_kernel", grp - 100 if grp >= 100 else 1)
        t_lr = timeit(lambda: lib.eyoc_spconv_build_local_rulebook(_lib.ctx(), tab, 27, n, _lib.ptr(local), _lib.ptr(ovf), _lib.stream_ptr()))
        run_st = lambda: _lib.check(lib.eyoc_spconv_staged(_lib.ctx(), tab, _lib.ptr(local), n, n_in, _lib.ptr(xs), cin, cin, _lib.ptr(wd), cout, None, None, 0, 0, _lib.ptr(out), cout, 1, _lib.ptr(osd), _lib.stream_ptr()))
        t = timeit(run_st)
        torch.cuda.synchronize()
        err = float((out - ref).abs().max()) / float(ref.abs().max())
        rec = local[: (n + 255) // 256 * 33408].view(-1, 33408).cpu().numpy()
        nu = rec[:, :4].copy().view(np.int32)[:, 0]
        msk = rec[:, 32784:32784 + 54].copy().view(np.uint16)[:, :27]
        nonempty = np.unpackbits(msk.view(np.uint8), axis=1).mean()
        res[grp] = (t, t_lr, err, nonempty, nu.mean(), nu.max())
    _lib.knob("eyoc_spconv_st_group_rows", 1); _lib.knob("eyoc_spconv_select_st_kernel", 1)
    print(f"lvl{lvl} {cin}->{cout} n={n} n_in={n_in} pairs={prs} ({prs / n:.2f}/row) overflow={int(ovf.item())} | gather kernel {t_ref:.3f} ms | " +
          "  ".join(f"staged group={g}: {v[0]:.3f} ms (records {v[1]:.3f} ms, rel err {v[2]:.1e}, non-empty blocks {v[3]:.3f}, distinct rows mean {v[4]:.0f} max {v[5]})" for g, v in res.items()), flush=True)

# ---- class-major order: fine rows sorted by parity class (= the class of their first valid offset), Z-order inside a class;
# the same staged kernel on the permuted table tells the distinct coarse rows per 256-row class tile and the non-empty blocks
print("class-major order:")
for lvl, cin, cout in ((0, 128, 64), (1, 256, 64), (2, 256, 128)):
    n = info["rows"][lvl]; n_in = info["rows"][lvl + 1]
    tab_t = torch.empty(27 * n, dtype=torch.int32, device="cuda")
    _lib.check(lib.eyoc_maps_copy_table(maps, 2, lvl, _lib.ptr(tab_t), _lib.stream_ptr()))
    T = tab_t.view(27, n)
    kk = torch.arange(27, device="cuda")
    cls_of_k = ((kk % 3 != 1).long() | ((kk // 3 % 3 != 1).long() << 1) | ((kk // 9 != 1).long() << 2))
    first = (T >= 0).float().argmax(dim=0)
    cls = cls_of_k[first]
    # every valid offset of a row must be of the row's class
    assert bool((((T >= 0) & (cls_of_k[:, None] != cls[None, :])).sum() == 0).item())
    perm = torch.sort(cls, stable=True).indices
    cnt = torch.bincount(cls, minlength=8).tolist()
    Tp = T[:, perm].contiguous()
    x = torch.randn(n_in, cin, device="cuda"); xs = torch.empty_like(x)
    lib.eyoc_split16_encode(_lib.ctx(), _lib.ptr(x), n_in, cin, cin, _lib.ptr(xs), cin, _lib.stream_ptr())
    W = np.random.default_rng(0).normal(size=(27, cin, cout)).astype(np.float32)
    packed = np.zeros(W.size, np.float32); osc = np.ones(1, np.float32)
    lib.eyoc_spconv_pack_weights_split16(W.ctypes.data, None, 27, cin, cout, packed.ctypes.data, osc.ctypes.data)
    wd = torch.from_numpy(packed).cuda(); osd = torch.from_numpy(osc).cuda()
    out = torch.empty(n, cout, device="cuda")
    local = torch.zeros(int(lib.eyoc_spconv_local_rulebook_bytes(n)), dtype=torch.uint8, device="cuda")
    ovf = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.knob("eyoc_spconv_st_group_rows", 1)
    lib.eyoc_spconv_build_local_rulebook(_lib.ctx(), _lib.ptr(Tp), 27, n, _lib.ptr(local), _lib.ptr(ovf), _lib.stream_ptr())
    res = {}
    for v in [1] + [v for v in variants if v != 1]:
        _lib.knob("eyoc_spconv_select_st_kernel", v)
        res[v] = timeit(lambda: _lib.check(lib.eyoc_spconv_staged(_lib.ctx(), _lib.ptr(Tp), _lib.ptr(local), n, n_in, _lib.ptr(xs), cin, cin, _lib.ptr(wd), cout, None, None, 0, 0, _lib.ptr(out), cout, 1, _lib.ptr(osd), _lib.stream_ptr())))
    _lib.knob("eyoc_spconv_select_st_kernel", 1)
    rec = local[: (n + 255) // 256 * 33408].view(-1, 33408).cpu().numpy()
    nu = rec[:, :4].copy().view(np.int32)[:, 0]
    msk = rec[:, 32784:32784 + 54].copy().view(np.uint16)[:, :27]
    nonempty = np.unpackbits(msk.view(np.uint8), axis=1).sum(axis=1) / 16.0      # offsets per chunk
    print(f"lvl{lvl}: class counts {cnt} overflow {int(ovf.item())} distinct rows mean {nu.mean():.0f} p90 {np.percentile(nu, 90):.0f} max {nu.max()} (two passes: {(nu > 639).mean():.3f}) "
          f"non-empty offsets per chunk {nonempty.mean():.2f} (ideal 3.375) | " + "  ".join(f"variant {v}: {t:.3f} ms" for v, t in res.items()), flush=True)
