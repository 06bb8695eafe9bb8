"""Phase timeline of the staged kernel's workgroups (needs the -DEYOC_ST_TRACE build: EYOC_HIP_LIB=.../libeyoc_hip_sttrace.so).
Per workgroup: start, header read, per 32-channel block {blob in, stage landed + barrier open, blob out}, end, and the CU it ran on."""
import ctypes as C, os, sys
import numpy as np, torch
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
lvl, cin, cout = int(os.environ.get("LVL", "0")), int(os.environ.get("CIN", "64")), int(os.environ.get("COUT", "64"))
n = info["rows"][lvl]
tab = lib.eyoc_maps_table(maps, 0, lvl)
x = torch.randn(n, cin, device="cuda"); xs = torch.empty_like(x)
lib.eyoc_split16_encode(_lib.ctx(), _lib.ptr(x), n, cin, cin, _lib.ptr(xs), cin, _lib.stream_ptr())
W = np.random.default_rng(0).normal(size=(27, cin, cout)).astype(np.float32)
packed = np.zeros(W.size, np.float32); osc = np.ones(1, np.float32)
lib.eyoc_spconv_pack_weights_split16(W.ctypes.data, None, 27, cin, cout, packed.ctypes.data, osc.ctypes.data)
wd = torch.from_numpy(packed).cuda(); osd = torch.from_numpy(osc).cuda()
out = torch.empty(n, cout, device="cuda")
local = torch.zeros(int(lib.eyoc_spconv_local_rulebook_bytes(n)), dtype=torch.uint8, device="cuda")
ovf = torch.zeros(1, dtype=torch.int32, device="cuda")
lib.eyoc_spconv_build_local_rulebook(_lib.ctx(), tab, 27, n, _lib.ptr(local), _lib.ptr(ovf), _lib.stream_ptr())
run = lambda: _lib.check(lib.eyoc_spconv_staged(_lib.ctx(), tab, _lib.ptr(local), n, n, _lib.ptr(xs), cin, cin, _lib.ptr(wd), cout, _lib.ptr(xs), _lib.ptr(xs), cin, 1, _lib.ptr(out), cout, 1, _lib.ptr(osd), _lib.stream_ptr()))
raw = C.CDLL(_lib.LIB_PATH)
raw.eyoc_debug_st_trace.argtypes = [C.c_void_p, C.c_size_t]
NT = 16
for _ in range(3): run()
torch.cuda.synchronize()
buf = np.zeros(16384 * NT, np.uint64)
raw.eyoc_debug_st_trace(buf.ctypes.data, buf.size)        # clears
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
raw.eyoc_debug_st_trace(buf.ctypes.data, buf.size)
t = buf.reshape(-1, NT).astype(np.int64)
t = t[t[:, 0] > 0]
hw = t[:, 9]; xcc = hw >> 32
# s_memrealtime: the constant 100 MHz counter (round 6; s_memtime's rate follows the clock and round 5's calibration on XCD-local spans
# came out inconsistent) - 10 ns resolution, phases are microseconds
spans = [t[xcc == x][:, 8].max() - t[xcc == x][:, 0].min() for x in np.unique(xcc)]
tick_us = 1.0 / 100.0
print(f"{len(t)} workgroups, kernel {ms * 1e3:.1f} us by events, XCD-local spans {min(spans) * tick_us:.0f}..{max(spans) * tick_us:.0f} us (100 ticks/us)")
d = lambda a, b: (t[:, a] - t[:, b]) * tick_us
names = [("launch -> first block", 2, 0), ("block0: stage wait + barrier", 3, 2), ("block0: offset loop", 4, 3), ("between blocks", 5, 4),
         ("block1: stage wait + barrier", 6, 5), ("block1: offset loop", 7, 6), ("epilogue", 8, 7), ("whole workgroup", 8, 0)]
for nm, a, b in names:
    v = d(a, b)
    print(f"  {nm:32s} mean {v.mean():7.2f} us  p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f}")
for b in (0, 1):          # thirds of an offset loop: barrier -> offset 9 -> offset 18 -> end
    a0, a1, a2, a3 = t[:, 3 + 3 * b], t[:, 10 + 2 * b], t[:, 11 + 2 * b], t[:, 4 + 3 * b]
    print(f"  block{b} loop by thirds (offsets 0-8, 9-17, 18-26): {((a1 - a0) * tick_us).mean():6.2f} {((a2 - a1) * tick_us).mean():6.2f} {((a3 - a2) * tick_us).mean():6.2f} us")
cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (xcc << 7)   # HW_ID: CU_ID [11:8], SE_ID [15:13]
ids = np.unique(cu)
life = (t[:, 8] - t[:, 0]) * tick_us
busy = np.array([life[cu == c].sum() for c in ids])
print(f"  distinct (xcc, se, cu) ids {len(ids)}; workgroups per id: mean {len(t) / len(ids):.1f}")
print(f"  sum of workgroup lifetimes per CU / (2 slots x kernel time): mean {busy.mean() / (2 * ms * 1e3):.2f}  min {busy.min() / (2 * ms * 1e3):.2f}  max {busy.max() / (2 * ms * 1e3):.2f}")
# concurrency on one CU: how often are both of its workgroups inside an offset loop at the same time
c0 = ids[0]; m = cu == c0
base = t[m][:, 0].min()
ev = []
for row in t[m]:
    ev += [(row[3] - base, +1), (row[4] - base, -1), (row[6] - base, +1), (row[7] - base, -1)]
ev.sort(); lvl = 0; last = 0; acc = [0, 0, 0]
for tt, dlt in ev:
    acc[min(lvl, 2)] += tt - last; last = tt; lvl += dlt
tot = sum(acc)
# how much of a workgroup's FIRST / SECOND loop runs while the CU's other workgroup is inside a loop too (all CUs)
ov = [[], []]
for c in ids:
    rows = t[cu == c]
    loops = [(r[3], r[4]) for r in rows] + [(r[6], r[7]) for r in rows]
    for r in rows:
        for which, (a, b) in enumerate(((r[3], r[4]), (r[6], r[7]))):
            if b <= a: continue
            o = sum(max(0, min(b, y) - max(a, x)) for (x, y) in loops if not (x == a and y == b))
            ov[which].append(o / (b - a))
print(f"  share of a loop spent beside another workgroup's loop on the same CU: first loop {np.mean(ov[0]):.2f}, second loop {np.mean(ov[1]):.2f}")
print(f"  CU {c0}: time with 0 / 1 / 2 workgroups inside an offset loop: {acc[0] / tot:.2f} / {acc[1] / tot:.2f} / {acc[2] / tot:.2f} of {tot * tick_us:.0f} us")
