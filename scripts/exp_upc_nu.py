"""Distinct coarse rows per class tile (spconv_upc.hip records) on the bench geometry, by class."""
import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import eyoc_amd, bench
from eyoc_amd import _lib, synthetic as syn
from test_gpu_split16 import morton_order
ps = bench.make_pairs(list(range(int(os.environ.get("PAIRS", "16")))))
clouds = []
for p in ps: clouds += [p["coords0"], p["coords1"]]
coords = syn.batch_coords(clouds); coords = coords[morton_order(coords)]
lib = _lib.load()
cm = eyoc_amd.CoordinateManager(torch.from_numpy(coords).cuda()); maps = cm.maps(); info = cm.info()
for lvl in (0, 1, 2):
    n = info["rows"][lvl]
    tab = lib.eyoc_maps_table(maps, 2, lvl)
    ws = torch.zeros(int(lib.eyoc_spconv_upc_bytes(n)) + 256, dtype=torch.uint8, device="cuda"); al = (ws.data_ptr() + 255) & ~255
    hi = np.zeros(19, np.int32)
    _lib.check(lib.eyoc_spconv_upc_build(_lib.ctx(), tab, n, C.c_void_p(al), hi.ctypes.data, _lib.stream_ptr()))
    raw = ws.cpu().numpy()[al - ws.data_ptr():]
    mt = (n + 127) // 128 + 8; nt = int(hi[0])
    rec0 = 256 + (mt * 4 + 255) // 256 * 256
    recs = raw[rec0:rec0 + nt * 14464].reshape(nt, 14464)
    hdr = recs[:, :8].copy().view(np.int32)
    nu, cls = hdr[:, 0], hdr[:, 1]
    msk = recs[:, 13328:13328 + 16].copy().view(np.uint16)
    for b in range(8):
        s = cls == b
        nz = np.unpackbits(msk[s].view(np.uint8), axis=1).sum(axis=1) / 16.0
        print(f"lvl{lvl} class {b} ({bin(b).count('1')} odd axes): {s.sum()} tiles, distinct rows mean {nu[s].mean():.0f} p50 {np.percentile(nu[s], 50):.0f} p90 {np.percentile(nu[s], 90):.0f} max {nu[s].max()}, <= 319: {(nu[s] <= 319).mean():.2f}, <= 212: {(nu[s] <= 212).mean():.2f}; non-empty offsets per chunk (pass 0) {nz.mean():.2f}")
