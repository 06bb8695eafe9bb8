"""Stride-1 staged records: how many tiles take two passes, and how many (chunk, offset) blocks they multiply twice."""
import os, sys, numpy as np, torch
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
REC, MASK_OFF = 33408, 32784
for lvl in range(4):
    n = info["rows"][lvl]
    tab = lib.eyoc_maps_table(maps, 0, lvl)
    local = torch.zeros(int(lib.eyoc_spconv_local_rulebook_bytes(n)), dtype=torch.uint8, device="cuda")
    ovf = torch.zeros(1, dtype=torch.int32, device="cuda")
    lib.eyoc_spconv_build_local_rulebook(_lib.ctx(), tab, 27, n, _lib.ptr(local), _lib.ptr(ovf), _lib.stream_ptr())
    nt = (n + 255) // 256
    rec = local[: nt * REC].view(nt, REC).cpu().numpy()
    nu = rec[:, :4].copy().view(np.int32)[:, 0]
    m = rec[:, MASK_OFF:MASK_OFF + 112].copy().view(np.uint16).reshape(nt, 2, 28)[:, :, :27]
    pc = lambda a: np.unpackbits(a.view(np.uint8), axis=-1).sum()
    two = nu > 639
    m1 = m[:, 1].copy(); m1[~two] = 0
    now = pc(m[:, 0]) + pc(m1)
    ideal = pc(m[:, 0] | m1)
    print(f"lvl{lvl}: {nt} tiles, two passes {two.mean():.3f} (distinct rows mean {nu.mean():.0f} p90 {np.percentile(nu, 90):.0f} max {nu.max()}), blocks multiplied {now} vs {ideal} if every block ran once ({now / ideal:.3f}x)")
