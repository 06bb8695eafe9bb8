"""Experiment: what does the staged kernel gain when whole (chunk, offset) blocks become empty?  The real level-0 table with the
neighbours at `drop` of the 27 offsets removed (every block of those offsets is then skipped by the loop's scalar branches; the
operand reads, the stage and everything else stay) - the time per removed MFMA tells how much of the kernel is matrix-issue time."""
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
coords = syn.batch_coords(clouds); coords = coords[morton_order(coords)]
cm = eyoc_amd.CoordinateManager(torch.from_numpy(coords).cuda())
maps = cm.maps(); lib = _lib.load(); info = cm.info()
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
for lvl, cin, cout in ((0, 64, 64), (2, 128, 128)):
    n = info["rows"][lvl]
    full = torch.empty((27, n), dtype=torch.int32, device="cuda")
    _lib.check(lib.eyoc_maps_copy_table(maps, 0, lvl, _lib.ptr(full), _lib.stream_ptr()))
    x = torch.randn(n, cin, device="cuda"); xs = torch.empty_like(x)
    lib.eyoc_split16_encode(_lib.ctx(), _lib.ptr(x), n, cin, cin, _lib.ptr(xs), cin, _lib.stream_ptr())
    W = np.random.default_rng(0).normal(size=(27, cin, cout)).astype(np.float32)
    packed = np.zeros(W.size, np.float32); osc = np.ones(1, np.float32)
    lib.eyoc_spconv_pack_weights_split16(W.ctypes.data, None, 27, cin, cout, packed.ctypes.data, osc.ctypes.data)
    wd = torch.from_numpy(packed).cuda(); osd = torch.from_numpy(osc).cuda()
    out = torch.empty(n, cout, device="cuda")
    order = np.random.default_rng(1).permutation([k for k in range(27) if k != 13])
    for drop in (0, 6, 13, 20, 26):
        tab = full.clone()
        if drop: tab[torch.from_numpy(order[:drop].copy()).cuda().long()] = -1
        pr = int((tab >= 0).sum())
        local = torch.zeros(int(lib.eyoc_spconv_local_rulebook_bytes(n)), dtype=torch.uint8, device="cuda")
        ovf = torch.zeros(1, dtype=torch.int32, device="cuda")
        lib.eyoc_spconv_build_local_rulebook(_lib.ctx(), _lib.ptr(tab), 27, n, _lib.ptr(local), _lib.ptr(ovf), _lib.stream_ptr())
        # non-empty (16-row chunk, offset) blocks
        v = (tab >= 0)
        padn = (-n) % 16
        ne = float(torch.nn.functional.pad(v, (0, padn)).reshape(27, -1, 16).any(2).float().mean())
        ts = []
        for rnd in range(3):
            ts.append(timeit(lambda: _lib.check(lib.eyoc_spconv_staged(_lib.ctx(), _lib.ptr(tab), _lib.ptr(local), n, n, _lib.ptr(xs), cin, cin, _lib.ptr(wd), cout, None, None, 0, 0, _lib.ptr(out), cout, 1, _lib.ptr(osd), _lib.stream_ptr()))))
        print(f"lvl{lvl} {cin}->{cout}: {drop:2d} offsets removed: pairs {pr} non-empty blocks {ne:.3f}  {np.median(ts):.3f} ms", flush=True)
# ---- second experiment: same table, same stage; a random fraction of the occupancy-mask bits cleared in the records (results wrong,
# timing valid): the pure price of the skipped MFMA blocks
print("mask-bit experiment (level-0 64->64)")
lvl, cin, cout = 0, 64, 64
n = info["rows"][lvl]
full = torch.empty((27, n), dtype=torch.int32, device="cuda")
_lib.check(lib.eyoc_maps_copy_table(maps, 0, lvl, _lib.ptr(full), _lib.stream_ptr()))
x = torch.randn(n, cin, device="cuda"); xs = torch.empty_like(x)
lib.eyoc_split16_encode(_lib.ctx(), _lib.ptr(x), n, cin, cin, _lib.ptr(xs), cin, _lib.stream_ptr())
W = np.random.default_rng(0).normal(size=(27, cin, cout)).astype(np.float32)
packed = np.zeros(W.size, np.float32); osc = np.ones(1, np.float32)
lib.eyoc_spconv_pack_weights_split16(W.ctypes.data, None, 27, cin, cout, packed.ctypes.data, osc.ctypes.data)
wd = torch.from_numpy(packed).cuda(); osd = torch.from_numpy(osc).cuda(); out = torch.empty(n, cout, device="cuda")
local0 = torch.zeros(int(lib.eyoc_spconv_local_rulebook_bytes(n)), dtype=torch.uint8, device="cuda")
ovf = torch.zeros(1, dtype=torch.int32, device="cuda")
lib.eyoc_spconv_build_local_rulebook(_lib.ctx(), _lib.ptr(full), 27, n, _lib.ptr(local0), _lib.ptr(ovf), _lib.stream_ptr())
REC, MASK_OFF = 33408, 32784
nt = (n + 255) // 256
g = torch.Generator(device="cuda").manual_seed(0)
for keep in (1.0, 0.8, 0.6, 0.4, 0.2, 0.0):
    local = local0.clone()
    rec = local[:nt * REC].view(nt, REC)
    m = rec[:, MASK_OFF:MASK_OFF + 112].contiguous().view(torch.int16).to(torch.int32) & 0xFFFF          # [nt, 56] 16-bit masks
    bits = (torch.rand((nt, 56, 16), device="cuda", generator=g) < keep)
    keepmask = (bits.to(torch.int32) << torch.arange(16, device="cuda", dtype=torch.int32)).sum(-1)
    m2 = (m & keepmask)
    setbits = float(sum(((m2 >> b) & 1).sum() for b in range(16))) / (nt * 27 * 16)
    rec[:, MASK_OFF:MASK_OFF + 112] = m2.to(torch.int16).view(torch.uint8).view(nt, 112)
    ts = [timeit(lambda: _lib.check(lib.eyoc_spconv_staged(_lib.ctx(), _lib.ptr(full), _lib.ptr(local), n, n, _lib.ptr(xs), cin, cin, _lib.ptr(wd), cout, None, None, 0, 0, _lib.ptr(out), cout, 1, _lib.ptr(osd), _lib.stream_ptr()))) for _ in range(3)]
    print(f"  mask bits kept {keep:.1f}: non-empty blocks {setbits:.3f}  {np.median(ts):.3f} ms", flush=True)
