"""Class-major transposed convolution (eyoc_amd/csrc/spconv_upc.hip; model/resunet.py:83-116 ME.MinkowskiConvolutionTranspose
kernel_size 3, stride 2): the partition of the fine rows by parity class, the records, and the layer against an fp64 product
and the gathering kernel."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

K = 27
CLASS_OF = np.array([(k % 3 != 1) | ((k // 3 % 3 != 1) << 1) | ((k // 9 != 1) << 2) for k in range(K)])
ORDER = np.concatenate([np.nonzero(CLASS_OF == b)[0] for b in range(8)])
START = np.concatenate([[0], np.cumsum([(CLASS_OF == b).sum() for b in range(8)])])
LR, U_OFF, LOC_OFF, MASK_OFF, OROW_OFF, HDR = 14464, 16, 5136, 13328, 13360, 256


def _maps(coords):
    import eyoc_amd
    from eyoc_amd import _lib as L
    from test_gpu_split16 import morton_order
    coords = coords[morton_order(coords)]
    cm = eyoc_amd.CoordinateManager(torch.from_numpy(coords).cuda())
    return cm, cm.maps(), cm.info(), L.load()


def _table(lib, maps, lvl, n):
    from eyoc_amd import _lib as L
    T = torch.empty(K * n, dtype=torch.int32, device="cuda")
    L.check(lib.eyoc_maps_copy_table(maps, 2, lvl, L.ptr(T), L.stream_ptr()))
    return T.view(K, n)


def _build(lib, tab, n):
    from eyoc_amd import _lib as L
    ws = torch.zeros(int(lib.eyoc_spconv_upc_bytes(n)) + 256, dtype=torch.uint8, device="cuda")
    al = (ws.data_ptr() + 255) & ~255
    info = np.zeros(19, np.int32)
    L.check(lib.eyoc_spconv_upc_build(L.ctx(), tab, n, C.c_void_p(al), info.ctypes.data, L.stream_ptr()))
    return ws, al, info


def _clouds(n_clouds, seed0=0):
    from eyoc_amd import synthetic as syn
    cl = []
    for i in range((n_clouds + 1) // 2):
        p = syn.make_pair(seed0 + i)
        cl += [p["coords0"], p["coords1"]]
    return syn.batch_coords(cl[:n_clouds])


def test_partition_and_records():
    """Every fine row sits in exactly one slot of a tile of its class; a slot's entries name the row's parents; the launch order
    is a permutation of the tiles that interleaves the classes along the Morton curve."""
    cm, maps, info, lib = _maps(_clouds(2))
    for lvl in (0, 1, 2):
        n = info["rows"][lvl]
        tab = lib.eyoc_maps_table(maps, 2, lvl)
        T = _table(lib, maps, lvl, n).cpu().numpy()
        ws, al, hi = _build(lib, tab, n)
        n_tiles, tstart, count, overflow = int(hi[0]), hi[1:10], hi[10:18], int(hi[18])
        valid = T >= 0
        cls = CLASS_OF[valid.argmax(axis=0)]
        assert overflow == 0 and (np.bincount(cls, minlength=8) == count).all() and count.sum() == n
        R = np.array([256] * 7 + [192])                                     # rows per tile: the 8-offset class takes 192
        assert (np.diff(tstart) == (count + R - 1) // R).all() and tstart[8] == n_tiles
        # a row's valid offsets all belong to its class
        assert not (valid & (CLASS_OF[:, None] != cls[None, :])).any()
        raw = ws.cpu().numpy()[al - ws.data_ptr():]
        max_tiles = (n + 127) // 128 + 8
        order = raw[HDR:HDR + 4 * max_tiles].view(np.int32)[:n_tiles]
        assert sorted(order.tolist()) == list(range(n_tiles))
        # along the launch order the classes alternate: any 16 consecutive entries hold at least 6 different classes (8 non-empty classes)
        tcls = np.searchsorted(tstart, order, side="right") - 1
        if n_tiles >= 64 and (count > 0).all():
            assert min(len(set(tcls[i:i + 16])) for i in range(0, n_tiles - 16)) >= 6
        rec0 = HDR + (max_tiles * 4 + 255) // 256 * 256
        recs = raw[rec0:rec0 + n_tiles * LR].reshape(n_tiles, LR)
        seen = np.zeros(n, np.int32)
        for t in range(n_tiles):
            r = recs[t]
            n_u, b = r[:8].view(np.int32)
            assert b == np.searchsorted(tstart, t, side="right") - 1 and 0 < n_u <= 1278
            U = r[U_OFF:U_OFF + 4 * 1280].view(np.int32)
            orow = r[OROW_OFF:OROW_OFF + 1024].view(np.int32).reshape(64, 4)          # [(16 w + j)][c]
            loc = r[LOC_OFF:LOC_OFF + 2 * 8 * 64 * 8].view(np.uint16).reshape(2, 8, 64, 4)
            msk = r[MASK_OFF:MASK_OFF + 32].view(np.uint16).reshape(2, 8)
            rows = orow[orow >= 0]
            assert (cls[rows] == b).all() and len(rows) <= R[b]
            np.add.at(seen, rows, 1)
            nk = START[b + 1] - START[b]
            n_pass = 2 if n_u > 639 else 1
            for i in range(nk):
                want = T[ORDER[START[b] + i]][np.maximum(orow, 0)]
                want[orow < 0] = -1
                got = np.full((64, 4), -1, np.int64)
                for p in range(n_pass):
                    a = loc[p, i].astype(np.int64)
                    slot = a // 64                                                   # slot_addr(l) = 64 l + 16 ((l >> 2) & 3)
                    assert ((a - slot * 64) == ((slot >> 2) & 3) * 16).all()
                    here = slot != 639
                    assert (got[here] == -1).all()                                   # a parent is staged in one pass only
                    got[here] = U[p * 639 + slot[here]]
                    # occupancy mask: bit 4 w + c <=> some row of chunk (w, c) has a parent in this pass
                    occ = here.reshape(4, 16, 4).any(axis=1)                         # [w][c]
                    bits = sum(int(occ[w, c]) << (4 * w + c) for w in range(4) for c in range(4))
                    assert bits == msk[p, i]
                assert (got == want).all()
        assert (seen == 1).all()


@pytest.mark.parametrize("lvl,cin,cout", [(0, 128, 64), (1, 256, 64), (2, 256, 128), (1, 64, 64), (0, 32, 128)])
def test_layer_against_fp64_and_the_gathering_kernel(lvl, cin, cout):
    from eyoc_amd import _lib as L
    cm, maps, info, lib = _maps(_clouds(3, seed0=7))
    n, n_in = info["rows"][lvl], info["rows"][lvl + 1]
    tab = lib.eyoc_maps_table(maps, 2, lvl)
    T = _table(lib, maps, lvl, n)
    g = torch.Generator(device="cuda").manual_seed(lvl)
    x = torch.randn(n_in, cin, device="cuda", generator=g)
    xs = torch.empty_like(x)
    lib.eyoc_split16_encode(L.ctx(), L.ptr(x), n_in, cin, cin, L.ptr(xs), cin, L.stream_ptr())
    W = np.random.default_rng(lvl).normal(size=(K, cin, cout)).astype(np.float32)
    bias = np.random.default_rng(lvl + 9).normal(size=cout).astype(np.float32)
    packed = np.zeros(W.size, np.float32)
    osc = np.ones(1, np.float32)
    lib.eyoc_spconv_pack_weights_split16(W.ctypes.data, None, K, cin, cout, packed.ctypes.data, osc.ctypes.data)
    wd, osd, bd = torch.from_numpy(packed).cuda(), torch.from_numpy(osc).cuda(), torch.from_numpy(bias).cuda()
    ws, al, hi = _build(lib, tab, n)
    assert hi[18] == 0
    ref = torch.zeros(n, cout, dtype=torch.float64, device="cuda")
    Wd = torch.from_numpy(W).cuda().double()
    for k in range(K):
        idx = T[k].long()
        m = idx >= 0
        ref[m] += x[idx[m]].double() @ Wd[k]
    ref = torch.relu(ref + bd.double())
    scale = float(ref.abs().max())
    for out_split in (0, 1):
        ld = cout + 32                                                 # a wider row: the layer writes columns [0, cout) of a concat buffer
        out = torch.full((n, ld), float("nan"), device="cuda")
        L.check(lib.eyoc_spconv_upc(L.ctx(), tab, C.c_void_p(al), n, n_in, L.ptr(xs), cin, cin, L.ptr(wd), cout, L.ptr(bd), 1, L.ptr(out), ld,
                                    out_split, L.ptr(osd), L.stream_ptr()))
        gat = torch.full((n, ld), float("nan"), device="cuda")
        L.check(lib.eyoc_spconv_ex(L.ctx(), tab, K, n, n_in, L.ptr(xs), cin, cin, L.ptr(wd), cout, L.ptr(bd), None, 0, 1, L.ptr(gat), ld, 1, out_split,
                                   L.ptr(osd), L.stream_ptr()))
        torch.cuda.synchronize()
        assert bool(torch.isnan(out[:, cout:]).all()), "columns outside the layer were written"
        if out_split:
            dec, decg = torch.empty(n, cout, device="cuda"), torch.empty(n, cout, device="cuda")
            lib.eyoc_split16_decode(L.ctx(), L.ptr(out), n, cout, ld, L.ptr(dec), cout, L.stream_ptr())
            lib.eyoc_split16_decode(L.ctx(), L.ptr(gat), n, cout, ld, L.ptr(decg), cout, L.stream_ptr())
        else:
            dec, decg = out[:, :cout], gat[:, :cout]
        e = float((dec.double() - ref).abs().max()) / scale
        eg = float((decg.double() - ref).abs().max()) / scale
        print(f"lvl {lvl} {cin}->{cout} split-out {out_split}: class-major {e:.2e}, gathering kernel {eg:.2e} of the largest output")
        assert e < 2e-6 and e <= 3 * eg + 2e-7
