"""Map build alone (Z-ordered, PAIRS=64 pairs = the bench batch): wall time per build; under rocprofv3 --kernel-trace --stats the per-kernel picture."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import eyoc_amd
from eyoc_amd import synthetic as syn
clouds = []
for s in range(int(os.environ.get('PAIRS', '64'))):
    p = syn.make_pair(s); clouds += [p["coords0"], p["coords1"]]
coords = torch.from_numpy(syn.batch_coords(clouds)).cuda()
if "ST_GROUP" in os.environ:
    from eyoc_amd import _lib
    _lib.knob("eyoc_spconv_st_group_rows", int(os.environ["ST_GROUP"]))
if "EYOC_DOWN" in os.environ:
    from eyoc_amd import _lib
    _lib.knob("eyoc_spconv_select_down_kernel", int(os.environ["EYOC_DOWN"]))
if "EYOC_MAPS_LAZY" in os.environ:
    from eyoc_amd import _lib
    _lib.knob("eyoc_maps_lazy_tables", int(os.environ["EYOC_MAPS_LAZY"]))
for _ in range(3):
    cm = eyoc_amd.CoordinateManager(coords); cm.maps(-1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    cm = eyoc_amd.CoordinateManager(coords); cm.maps(-1)
torch.cuda.synchronize()
print("maps build ms", (time.perf_counter() - t0) / 10 * 1e3)
