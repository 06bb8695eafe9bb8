"""GPU parity of the split16 sparse convolution (three fp16 MFMAs per product on hi/lo-split operands,
eyoc_amd/csrc/spconv_wave.hip MATH = 1): the SPLIT16 row format, single layers against an fp64 restatement next to the
fp32-MFMA path, and the whole network against the oracle at the same 1e-4 bar as the fp32 path."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REL = 1e-4


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _lib():
    from eyoc_amd import _lib as L
    return L, L.load()


@pytest.fixture(params=[0, 2], ids=["wave-private", "row-stationary"], autouse=True)
def split16_kernel(request):
    """Every test runs on both split16 kernels (spconv_wave.hip MATH = 1 / spconv_rs.hip); production picks per layer."""
    L, lib = _lib()
    prev = L.knob("eyoc_spconv_select_split16_kernel", request.param)
    yield request.param
    L.knob("eyoc_spconv_select_split16_kernel", prev)


def encode(x):
    L, lib = _lib()
    x = x.contiguous()
    out = torch.empty_like(x)
    L.check(lib.eyoc_split16_encode(L.ctx(), L.ptr(x), x.shape[0], x.shape[1], x.stride(0), L.ptr(out), out.stride(0), L.stream_ptr()))
    return out


def decode(x):
    L, lib = _lib()
    out = torch.empty_like(x)
    L.check(lib.eyoc_split16_decode(L.ctx(), L.ptr(x), x.shape[0], x.shape[1], x.stride(0), L.ptr(out), out.stride(0), L.stream_ptr()))
    return out


def test_split16_row_format_round_trip():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(size=(500, 64)), rng.normal(size=(500, 64)) * 1e-3, rng.uniform(-6e4, 6e4, (24, 64)),
                        np.zeros((8, 64))]).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    enc = encode(xd)
    back = decode(enc).cpu().numpy()
    err = np.abs(back - x)
    bound = np.maximum(np.abs(x) * 2.0 ** -21, 2.0 ** -24)      # 22-bit significand, fp16 subnormal floor of the lo half
    assert (err <= bound).all(), float((err / bound).max())
    print(f"split16 round trip: max rel err {float((err / np.maximum(np.abs(x), 1e-30))[np.abs(x) > 0.125].max()):.2e}")
    # layout: per 32 channels 32 fp16 hi (64 bytes) then 32 fp16 lo (64 bytes)
    raw = enc.cpu().numpy().view(np.float16).reshape(len(x), 2, 2, 32)
    np.testing.assert_array_equal(raw[:, :, 0, :].reshape(len(x), 64), x.astype(np.float16))
    hi = raw[:, :, 0, :].reshape(len(x), 64).astype(np.float32)
    np.testing.assert_array_equal(raw[:, :, 1, :].reshape(len(x), 64), (x - hi).astype(np.float16))


def run_layer_split(nbr, x, W, bias=None, scale=None, res=None, relu=False, out_split=False):
    L, lib = _lib()
    K, cin, cout = W.shape
    packed = np.zeros(K * cin * cout, np.float32)
    os_ = np.zeros(1, np.float32)
    sc = None if scale is None else np.ascontiguousarray(scale, np.float32)
    assert lib.eyoc_spconv_pack_weights_split16(np.ascontiguousarray(W).ctypes.data, None if sc is None else sc.ctypes.data, K, cin,
                                                cout, packed.ctypes.data, os_.ctypes.data) == 0
    dev = torch.device("cuda")
    n_out = nbr.shape[1] if nbr is not None else x.shape[0]
    xin = encode(torch.from_numpy(x).to(dev))
    rin = None if res is None else encode(torch.from_numpy(res).to(dev))
    out = torch.full((n_out, cout), -555.0, device=dev)
    wd, osd = torch.from_numpy(packed).to(dev), torch.from_numpy(os_).to(dev)
    bd = None if bias is None else torch.from_numpy(np.ascontiguousarray(bias, np.float32)).to(dev)
    nd = None if nbr is None else torch.from_numpy(np.ascontiguousarray(nbr, np.int32)).to(dev)
    L.check(lib.eyoc_spconv_ex(L.ctx(), L.ptr(nd), K, n_out, x.shape[0], L.ptr(xin), xin.stride(0), cin, L.ptr(wd), cout, L.ptr(bd), L.ptr(rin),
                               0 if rin is None else rin.stride(0), 1 if relu else 0, L.ptr(out), out.stride(0), 1,
                               1 if out_split else 0, L.ptr(osd), L.stream_ptr()), "eyoc_spconv_ex")
    if out_split:
        out = decode(out)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def layer_f64(nbr, x, W, bias=None, scale=None, res=None, relu=False):
    """fp64 restatement of one layer (independent of both GPU paths' rounding)."""
    if nbr is None:
        nbr = np.arange(x.shape[0], dtype=np.int32)[None]
    Wd = W.astype(np.float64) * (1.0 if scale is None else scale.astype(np.float64)[None, None, :])
    out = np.zeros((nbr.shape[1], W.shape[2]))
    xd = x.astype(np.float64)
    for k in range(nbr.shape[0]):
        v = nbr[k] >= 0
        out[v] += xd[nbr[k][v]] @ Wd[k]
    if bias is not None:
        out += bias[None]
    if res is not None:
        out += res
    return np.maximum(out, 0) if relu else out


@pytest.fixture(scope="module")
def maps():
    from test_gpu_spconv import small_maps
    return small_maps(0, 4000)


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 64), (32, 64), (128, 128), (256, 64), (128, 256), (96, 64), (64, 32)])
def test_split16_layer_is_as_accurate_as_fp32_mfma(maps, cin, cout, split16_kernel):
    """Against an fp64 restatement: the split16 layer's error must be of the order of the fp32-MFMA layer's own
    (both ~1e-7 relative to the largest output), three orders of magnitude inside the 1e-4 bar."""
    from eyoc_amd import _lib as L
    from test_gpu_spconv import run_layer
    nbr = maps["s1"][0]
    rng = np.random.default_rng(cin * 7 + cout)
    x = np.abs(rng.normal(size=(nbr.shape[1], cin))).astype(np.float32)         # post-ReLU-like activations
    x[rng.random(x.shape) < 0.3] = 0
    W = (rng.normal(size=(27, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    s = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    r = rng.normal(size=(nbr.shape[1], cout)).astype(np.float32)
    want = layer_f64(nbr, x, W, bias=b, scale=s, res=r, relu=True)
    lib = L.load()
    prev = L.knob("eyoc_spconv_select_kernel", 1)
    try:
        got32 = run_layer(nbr, x, W, bias=b, scale=s, res=r, relu=True)
    finally:
        L.knob("eyoc_spconv_select_kernel", prev)
    got16 = run_layer_split(nbr, x, W, bias=b, scale=s, res=r, relu=True)
    got16s = run_layer_split(nbr, x, W, bias=b, scale=s, res=r, relu=True, out_split=True)
    e32, e16, e16s = rel_err(got32, want), rel_err(got16, want), rel_err(got16s, want)
    print(f"{cin}->{cout}: vs fp64  fp32-mfma {e32:.2e}  split16 {e16:.2e}  split16 + split store {e16s:.2e}")
    # the claim of DESIGN 3.2b, as worded there.  Wave-private kernel (per-offset partial sums): the split16 error IS the
    # fp32-MFMA path's (realised 1.2e-7 .. 1.9e-7 of the largest output for both).  Row-stationary kernel (one fp32 chain over all
    # 27 offsets per accumulator, 64-wide operand blocks): 2.8e-7 .. 1.1e-6, i.e. up to 6x the fp32 path's - still 100x inside
    # the 1e-4 bar, but NOT "equal", and the bound below says so.
    if split16_kernel == 0:
        assert e16 <= 2 * e32 + 1e-7 and e16s <= 2 * e32 + 1e-7, (e32, e16, e16s)
    else:
        assert e16 < 2e-6 and e16s < 2e-6 and e16 <= 8 * e32 + 2e-7, (e32, e16, e16s)


def test_split16_identity_strided_transposed_and_ragged(maps):
    rng = np.random.default_rng(3)
    n1, n2 = len(maps["cm"][0]), len(maps["cm"][1])
    x1 = rng.normal(size=(n1, 32)).astype(np.float32)
    Wd = (rng.normal(size=(27, 32, 64)) / 16).astype(np.float32)
    down = run_layer_split(maps["down"][0], x1, Wd)
    assert down.shape == (n2, 64) and rel_err(down, layer_f64(maps["down"][0], x1, Wd)) < 2e-6
    Wu = (rng.normal(size=(27, 64, 32)) / 16).astype(np.float32)
    assert rel_err(run_layer_split(maps["up"][0], down, Wu), layer_f64(maps["up"][0], down, Wu)) < 2e-6
    W1 = (rng.normal(size=(1, 96, 64)) / 10).astype(np.float32)
    x96 = rng.normal(size=(n1, 96)).astype(np.float32)
    assert rel_err(run_layer_split(None, x96, W1, relu=True), layer_f64(None, x96, W1, relu=True)) < 2e-6
    for n in (1, 17, 65, 200):
        from oracle import coords as oc
        c = np.unique(rng.integers(-6, 6, size=(4 * n + 8, 3)), axis=0)[:n]
        cm = oc.CoordMap(np.concatenate([np.zeros((len(c), 1), np.int64), c], 1), 1)
        nbr = oc.kernel_map(cm, cm, 3)
        x = rng.normal(size=(len(c), 32)).astype(np.float32)
        W = (rng.normal(size=(27, 32, 32)) / 10).astype(np.float32)
        assert rel_err(run_layer_split(nbr, x, W), layer_f64(nbr, x, W)) < 2e-6


def test_split16_forward_error_growth_against_fp64():
    """Error growth over the 23 layers: the whole forward of a ~5k-voxel cloud in both arithmetics against the oracle run
    in float64 (oracle/resunet.py dtype=torch.float64).  split16 must not be worse than twice the fp32-MFMA path."""
    from eyoc_amd import synthetic as syn
    from oracle import resunet as orr
    from test_gpu_round2 import _model
    p = syn.make_pair(5, beams=24, azimuths=700, band=None)
    coords = syn.batch_coords([p["coords0"]])
    assert 3000 < len(coords) < 12000, len(coords)
    model, sd = _model()
    want = orr.resunet_forward(sd, coords, p["feats0"], dtype=torch.float64).numpy()
    errs = {}
    for mode in ("fp32", "split16"):
        model.spconv_math = mode
        got = _forward(model, coords, p["feats0"]).astype(np.float64)
        assert model.last_spconv_math == mode
        errs[mode] = float(np.abs(got - want).max() / np.abs(want).max())
    model.spconv_math = "auto"
    print(f"forward vs fp64 oracle ({len(coords)} voxels): fp32-mfma {errs['fp32']:.2e}  split16 {errs['split16']:.2e}")
    assert errs["fp32"] < 1e-5 and errs["split16"] <= 2 * errs["fp32"] + 1e-7, errs


def _forward(model, coords, feats):
    import eyoc_amd
    return model(eyoc_amd.SparseTensor(torch.from_numpy(feats).cuda(), coordinates=torch.from_numpy(coords).cuda())).F.cpu().numpy()


def test_split16_forward_matches_oracle_like_fp32(request):
    """Whole ResUNetBN2C forward on a 31k-voxel cloud in both arithmetics against the CPU oracle, same bar (1e-4 of the
    largest feature, cosine 1 - 1e-6 per row); split16 is selected explicitly and must really have run."""
    from eyoc_amd import synthetic as syn
    from oracle import resunet as orr
    from test_gpu_round2 import _model
    p = syn.make_pair(1)
    coords = syn.batch_coords([p["coords0"]])
    model, sd = _model()
    want = orr.resunet_forward(sd, coords, p["feats0"]).numpy()
    errs = {}
    for mode in ("fp32", "split16"):
        model.spconv_math = mode
        got = _forward(model, coords, p["feats0"])
        assert model.last_spconv_math == mode
        errs[mode] = rel_err(got, want)
        cos = (got * want).sum(1)
        assert errs[mode] < REL and cos.min() > 1 - 1e-6, (mode, errs[mode], float(cos.min()))
        np.testing.assert_allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    print(f"forward vs oracle: fp32-mfma {errs['fp32']:.2e}  split16 {errs['split16']:.2e}")
    # batched small clouds + permuted rows: the row-order contract holds in split16 too
    rng = np.random.default_rng(11)
    c = np.unique(rng.integers(-12, 12, size=(3000, 3)) // np.array([1, 1, 2]), axis=0)
    rng.shuffle(c)
    cb = syn.batch_coords([c[:900].astype(np.int32), c[900:].astype(np.int32)])
    ones = np.ones((len(cb), 1), np.float32)
    a = _forward(model, cb, ones)
    assert rel_err(a, orr.resunet_forward(sd, cb, ones).numpy()) < REL
    perm = rng.permutation(len(cb))
    assert rel_err(_forward(model, cb[perm], ones), a[perm]) < 1e-5
    # automatic mode: small batches stay on fp32
    model.spconv_math = "auto"
    _forward(model, cb, ones)
    assert model.last_spconv_math == "fp32"


def test_forward_on_z_ordered_rows_with_the_staged_kernel(request):
    """Production path of the large batches, forced onto a 2 x 31k-voxel batch: rows kept in Z-order inside the maps,
    conv1 reading and the last layer writing through the permutation, the 64+-channel stride-1 layers on the tile-local
    input stage.  The caller sees its own row order; same bar as every other forward."""
    from eyoc_amd import _lib as L, synthetic as syn
    from oracle import resunet as orr
    from test_gpu_round2 import _model
    lib = L.load()
    p = syn.make_pair(4)
    coords = syn.batch_coords([p["coords0"], p["coords1"]])
    feats = np.concatenate([p["feats0"], p["feats1"]])
    model, sd = _model()
    want = orr.resunet_forward(sd, coords, feats).numpy()
    prev = L.knob("eyoc_maps_internal_order", 1) - 2
    prev_up = L.knob("eyoc_spconv_select_up_kernel", -1)
    try:
        for mode, up in (("split16", 0), ("split16", 1), ("split16", 2), ("fp32", 0)):   # up = 1: transposed convolutions on spconv_up.hip, 2: spconv_upc.hip
            L.knob("eyoc_spconv_select_up_kernel", up)
            prev_min = L.knob("eyoc_spconv_upc_min_rows", 0)        # class-major tiles whatever the size of the batch
            model.spconv_math = mode
            got = _forward(model, coords, feats)
            assert model.last_spconv_math == mode
            e = rel_err(got, want)
            cos = (got * want).sum(1)
            print(f"z-ordered forward, {mode}, staged transposed convolutions {up}: err {e:.2e}")
            L.knob("eyoc_spconv_upc_min_rows", prev_min)
            assert e < REL and cos.min() > 1 - 1e-6, (mode, e, float(cos.min()))
        L.knob("eyoc_spconv_select_up_kernel", prev_up)
        # the permutation is invisible: permuting the caller's rows permutes the output
        rng = np.random.default_rng(3)
        perm = rng.permutation(len(coords))
        model.spconv_math = "split16"
        a = _forward(model, coords, feats)
        assert rel_err(_forward(model, coords[perm], feats[perm]), a[perm]) < 1e-5
    finally:
        L.knob("eyoc_maps_internal_order", prev)
        L.knob("eyoc_spconv_select_up_kernel", prev_up)
        model.spconv_math = "auto"


def test_split16_other_channel_tables():
    from eyoc_amd import synthetic as syn
    from oracle import resunet as orr
    import eyoc_amd
    rng = np.random.default_rng(12)
    c = np.unique(rng.integers(-10, 10, size=(1500, 3)), axis=0).astype(np.int32)
    coords = syn.batch_coords([c])
    sd = syn.make_weights(seed=5, in_channels=3, conv1_kernel_size=3, tr_channels=(None, 64, 64, 64, 64))
    m = eyoc_amd.load_model("ResUNetBN2B")(3, 32, bn_momentum=0.05, conv1_kernel_size=3, normalize_feature=False)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m = m.cuda().eval()
    m.spconv_math = "split16"
    f3 = rng.normal(size=(len(coords), 3)).astype(np.float32)
    got = _forward(m, coords, f3)
    want = orr.resunet_forward(sd, coords, f3, normalize_feature=False, conv1_kernel_size=3).numpy()
    assert m.last_spconv_math == "split16" and rel_err(got, want) < REL


# ------------------------------------------------------------------------------------------------ staged kernel
def morton_order(coords):
    """Stable Z-order of (batch, x, y, z) rows: batch outermost, 18 interleaved bits per axis (the order eyoc_maps_build
    stores a level's rows in for large batches)."""
    c = coords.astype(np.int64)
    b, x, y, z = c[:, 0], c[:, 1] + (1 << 17), c[:, 2] + (1 << 17), c[:, 3] + (1 << 17)

    def spread(v):
        o = np.zeros_like(v)
        for i in range(18):
            o |= ((v >> i) & 1) << (3 * i)
        return o
    return np.argsort((b << 54) | spread(x) | (spread(y) << 1) | (spread(z) << 2), kind="stable")


def run_layer_staged(nbr, x, W, bias=None, scale=None, res=None, relu=False, out_split=True):
    L, lib = _lib()
    K, cin, cout = W.shape
    packed = np.zeros(K * cin * cout, np.float32)
    os_ = np.zeros(1, np.float32)
    sc = None if scale is None else np.ascontiguousarray(scale, np.float32)
    assert lib.eyoc_spconv_pack_weights_split16(np.ascontiguousarray(W).ctypes.data, None if sc is None else sc.ctypes.data, K, cin,
                                                cout, packed.ctypes.data, os_.ctypes.data) == 0
    dev = torch.device("cuda")
    n_out = nbr.shape[1]
    nd = torch.from_numpy(np.ascontiguousarray(nbr, np.int32)).to(dev)
    local = torch.zeros(int(lib.eyoc_spconv_local_rulebook_bytes(n_out)), dtype=torch.uint8, device=dev)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    L.check(lib.eyoc_spconv_build_local_rulebook(L.ctx(), L.ptr(nd), K, n_out, L.ptr(local), L.ptr(ovf), L.stream_ptr()))
    assert int(ovf.item()) == 0, "a tile has more distinct input rows than two passes stage"
    xin = encode(torch.from_numpy(x).to(dev))
    rin = None if res is None else encode(torch.from_numpy(res).to(dev))
    out = torch.full((n_out, cout), -555.0, device=dev)
    wd, osd = torch.from_numpy(packed).to(dev), torch.from_numpy(os_).to(dev)
    bd = None if bias is None else torch.from_numpy(np.ascontiguousarray(bias, np.float32)).to(dev)
    L.check(lib.eyoc_spconv_staged(L.ctx(), L.ptr(nd), L.ptr(local), n_out, x.shape[0], L.ptr(xin), xin.stride(0), cin, L.ptr(wd), cout,
                                   L.ptr(bd), L.ptr(rin), 0 if rin is None else rin.stride(0), 1 if relu else 0, L.ptr(out),
                                   out.stride(0), 1 if out_split else 0, L.ptr(osd), L.stream_ptr()), "eyoc_spconv_staged")
    if out_split:
        out = decode(out)
    torch.cuda.synchronize()
    return out.cpu().numpy(), local


@pytest.fixture(scope="module")
def morton_maps():
    from eyoc_amd import synthetic as syn
    from oracle import coords as oc
    p = syn.make_pair(2, beams=32, azimuths=1000, band=None)
    coords = syn.batch_coords([p["coords0"], p["coords1"]])
    return oc.build_maps(coords[morton_order(coords)])


@pytest.mark.parametrize("cin,cout,level", [(64, 64, 0), (32, 32, 0), (128, 128, 1), (256, 256, 2), (64, 64, 1)])
def test_staged_kernel_vs_fp64(morton_maps, cin, cout, level):
    """Tile-local input stage (spconv_st.hip) on Morton-ordered rows: local rulebooks never overflow, every distinct input
    row is staged once per 32-channel block, results at the split16 accuracy against the fp64 restatement."""
    nbr = morton_maps["s1"][level]
    n = nbr.shape[1]
    rng = np.random.default_rng(cin + cout + level)
    x = np.abs(rng.normal(size=(n, cin))).astype(np.float32)
    W = (rng.normal(size=(27, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    s = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    r = rng.normal(size=(n, cout)).astype(np.float32)
    want = layer_f64(nbr, x, W, bias=b, scale=s, res=r, relu=True)
    got, local = run_layer_staged(nbr, x, W, bias=b, scale=s, res=r, relu=True)
    got32, _ = run_layer_staged(nbr, x, W, bias=b, scale=s, res=r, relu=True, out_split=False)
    e, e32 = rel_err(got, want), rel_err(got32, want)
    # the implementations of the offset loop (assembly with / without empty-block branches,
    # compiler-scheduled C++), in 64- and in 32-channel workgroups,
    # multiply the same products in the same order: bit-identical outputs
    L, lib = _lib()
    prev, prev_split = L.knob("eyoc_spconv_select_st_kernel", -1), L.knob("eyoc_spconv_st_split_below", -1)
    try:
        for split in (prev_split, 0):                      # a cloud this small takes 32-channel workgroups; 0: the wide kernels
            L.knob("eyoc_spconv_st_split_below", split)
            for variant in (0, 1, 2):
                L.knob("eyoc_spconv_select_st_kernel", variant)
                alt, _ = run_layer_staged(nbr, x, W, bias=b, scale=s, res=r, relu=True, out_split=False)
                np.testing.assert_array_equal(alt, got32, err_msg=f"staged kernel variant {variant}, split below {split}")
    finally:
        L.knob("eyoc_spconv_select_st_kernel", prev)
        L.knob("eyoc_spconv_st_split_below", prev_split)
    # the local rulebook: one record per 256-row tile - n_unique first, then the row list, the slot entries and, last, per pass
    # 28 16-bit occupancy masks (bit 4 w + c of mask k: some row of rows 64 w + 16 c .. + 15 has a neighbour at offset k)
    # ... and, last, the tile's row map: slot 64 w + 16 c + j holds local row rowmap[(16 w + j) * 4 + c] (the builder groups a tile's rows
    # by neighbour pattern so that more (chunk, offset) blocks come out empty; in row order with eyoc_spconv_st_group_rows(0))
    REC, MASK_OFF, RM_OFF = 33408, 32784, 32896
    n_tiles = (n + 255) // 256
    lr = local.cpu().numpy()[:n_tiles * REC].reshape(-1, REC)
    n_u = lr[:, :4].copy().view(np.int32)[:, 0]
    masks = lr[:, MASK_OFF:MASK_OFF + 56].copy().view(np.uint16)[:, :27]
    rowmap = lr[:, RM_OFF:RM_OFF + 256].reshape(n_tiles, 4, 16, 4).transpose(0, 1, 3, 2).reshape(n_tiles, 256).astype(np.int64)   # [tile, slot]
    assert (np.sort(rowmap, axis=1) == np.arange(256)).all()
    occ = np.zeros((n_tiles * 256, 27), bool)
    occ[:n] = (nbr >= 0).T
    occ_rows = occ.reshape(n_tiles, 256, 27)
    occ_slots = np.take_along_axis(occ_rows, rowmap[:, :, None], axis=1)
    want_masks = (occ_slots.reshape(n_tiles, 16, 16, 27).any(axis=2) * (1 << np.arange(16))[None, :, None]).sum(axis=1)
    f_rows, f_slots = occ_rows.reshape(n_tiles, 16, 16, 27).any(axis=2).mean(), occ_slots.reshape(n_tiles, 16, 16, 27).any(axis=2).mean()
    assert f_slots < f_rows - 0.03, (f_rows, f_slots)
    # grouping only re-orders a tile's rows inside its workgroup: bit-identical outputs
    prev_g = L.knob("eyoc_spconv_st_group_rows", 0)
    try:
        plain, local_plain = run_layer_staged(nbr, x, W, bias=b, scale=s, res=r, relu=True, out_split=False)
    finally:
        L.knob("eyoc_spconv_st_group_rows", prev_g)
    np.testing.assert_array_equal(plain, got32)
    rm_plain = local_plain.cpu().numpy()[:n_tiles * REC].reshape(-1, REC)[:, RM_OFF:RM_OFF + 256].reshape(n_tiles, 4, 16, 4).transpose(0, 1, 3, 2).reshape(n_tiles, 256)
    assert (rm_plain == np.arange(256)).all()
    single = n_u <= 639                                    # tiles staged in one pass: the first-pass masks are the whole story
    np.testing.assert_array_equal(masks[single], want_masks[single].astype(np.uint16))
    pairs = int((nbr >= 0).sum())
    print(f"staged {cin}->{cout} level {level}: n {n} err {e:.2e} / {e32:.2e}  distinct rows per tile mean {n_u.mean():.0f} max {n_u.max()}"
          f"  re-use {pairs / n_u.sum():.2f}x  non-empty (16-row chunk, offset) blocks {f_rows:.2f} in row order, {f_slots:.2f} grouped")
    assert e < 2e-6 and e32 < 2e-6 and n_u.max() <= 1278 and n_u.min() >= 1


def test_local_rulebook_counts_tiles_it_cannot_stage_instead_of_hanging():
    """Rows in NO spatial order: a 256-row tile of a dense grid then references up to 256 x 27 distinct rows - more than the
    1278 the stage takes in two passes, and more than the builder's 4096-slot LDS hash holds.  The builder gives up on such a
    tile (n_unique = -1), counts it in *overflow and terminates; tiles of the same cloud in Morton order build cleanly."""
    from oracle import coords as oc
    L, lib = _lib()
    g = np.stack(np.meshgrid(np.arange(26), np.arange(26), np.arange(26), indexing="ij"), -1).reshape(-1, 3)
    coords = np.concatenate([np.zeros((len(g), 1), np.int64), g], 1).astype(np.int32)
    dev = torch.device("cuda")
    REC = 33408
    for order, expect_overflow in ((np.random.default_rng(0).permutation(len(coords)), True), (morton_order(coords), False)):
        nbr = oc.build_maps(coords[order])["s1"][0]
        n = nbr.shape[1]
        nd = torch.from_numpy(np.ascontiguousarray(nbr, np.int32)).to(dev)
        local = torch.zeros(int(lib.eyoc_spconv_local_rulebook_bytes(n)), dtype=torch.uint8, device=dev)
        ovf = torch.zeros(1, dtype=torch.int32, device=dev)
        L.check(lib.eyoc_spconv_build_local_rulebook(L.ctx(), L.ptr(nd), 27, n, L.ptr(local), L.ptr(ovf), L.stream_ptr()))
        torch.cuda.synchronize()
        n_tiles = (n + 255) // 256
        n_u = local.cpu().numpy()[:n_tiles * REC].reshape(-1, REC)[:, :4].copy().view(np.int32)[:, 0]
        distinct = np.array([len(np.unique(t[t >= 0])) for t in np.array_split(nbr, np.arange(256, n, 256), axis=1)])
        if expect_overflow:
            assert distinct.max() > 4096                                   # the case the hash cannot hold at all
            assert int(ovf.item()) == int((distinct > 1278).sum()) > 0
            np.testing.assert_array_equal(n_u[distinct > 1278], -1)
        else:
            assert int(ovf.item()) == 0
        np.testing.assert_array_equal(n_u[distinct <= 1278], distinct[distinct <= 1278])


def test_staged_kernel_refuses_channel_widths_it_cannot_tile(morton_maps):
    """96 output channels are neither one 64-channel workgroup nor a power-of-two number of them: the staged entry point
    says so instead of computing two thirds of the layer (launch_spconv keeps such layers on the gathering kernels)."""
    nbr = morton_maps["s1"][2]
    rng = np.random.default_rng(1)
    x = rng.normal(size=(nbr.shape[1], 64)).astype(np.float32)
    W = (rng.normal(size=(27, 64, 96)) / 10).astype(np.float32)
    with pytest.raises(Exception):
        run_layer_staged(nbr, x, W)


def test_first_convolution_as_mfma_gemm_with_a_3x3x3_window():
    """conv1_mfma_kernel (C_in = 1, 32 output channels, split16 consumers) with the 27-position window; the bench's 5^3
    window is covered by every split16 forward above."""
    from eyoc_amd import synthetic as syn
    from oracle import resunet as orr
    import eyoc_amd
    p = syn.make_pair(6, beams=32, azimuths=1000, band=None)
    coords = syn.batch_coords([p["coords0"]])
    sd = syn.make_weights(seed=7, in_channels=1, conv1_kernel_size=3)
    m = eyoc_amd.load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, conv1_kernel_size=3, normalize_feature=True)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m = m.cuda().eval()
    m.spconv_math = "split16"
    feats = np.random.default_rng(2).uniform(0.5, 2.0, size=(len(coords), 1)).astype(np.float32)
    got = _forward(m, coords, feats)
    want = orr.resunet_forward(sd, coords, feats, conv1_kernel_size=3).numpy()
    assert m.last_spconv_math == "split16" and rel_err(got, want) < REL


@pytest.mark.parametrize("normalize", [True, False])
def test_fused_tail_matches_the_two_layer_tail_and_the_oracle(normalize):
    """conv1_tr -> ReLU -> final (+ bias) -> row normalisation in one kernel (spconv_tail.hip) against the same forward with
    the two layers as separate launches (the products are summed in another order: equal to a few fp32 ulps, not bitwise)
    and against the oracle; with and without the normalisation (model/resunet.py:187-191); ragged row count; planted zero row."""
    import eyoc_amd
    from eyoc_amd import synthetic as syn
    from oracle import resunet as orr
    L, lib = _lib()
    p = syn.make_pair(7, beams=24, azimuths=700, band=None)
    coords = syn.batch_coords([p["coords0"]])
    assert len(coords) % 16 != 0 or len(coords) > 0
    sd = syn.make_weights()
    model = eyoc_amd.load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, conv1_kernel_size=5, normalize_feature=normalize)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model = model.cuda().eval()
    model.spconv_math = "split16"
    want = orr.resunet_forward(sd, coords, p["feats0"], normalize_feature=normalize).numpy()
    outs = {}
    prev = L.knob("eyoc_model_fuse_tail", -1)
    prev_split = L.knob("eyoc_spconv_st_split_below", 0)        # 64-channel workgroups for this one cloud too: what carries the tail in mode 2
    try:
        for fuse in (0, 1, 2):
            L.knob("eyoc_model_fuse_tail", fuse)
            outs[fuse] = _forward(model, coords, p["feats0"])
            assert model.last_spconv_math == "split16"
        # mode 2 (round 6, the default): the tail in the epilogue of block2_tr.conv2 - the same products in the same order as the
        # tail kernel of mode 1 (the layer's output is encoded to split16 in registers instead of being stored and re-read)
        assert np.array_equal(outs[2], outs[1], equal_nan=True)
        L.knob("eyoc_spconv_st_split_below", prev_split)
        small = _forward(model, coords, p["feats0"])                   # 32-channel workgroups cannot carry it: mode 2 falls back to mode 1
        assert np.abs(small - outs[1]).max() / np.abs(want).max() < 2e-6
    finally:
        L.knob("eyoc_model_fuse_tail", prev)
        L.knob("eyoc_spconv_st_split_below", prev_split)
    scale = np.abs(want).max()
    e_fused, e_two = np.abs(outs[1] - want).max() / scale, np.abs(outs[0] - want).max() / scale
    d = np.abs(outs[1] - outs[0]).max() / scale
    print(f"tail (normalize={normalize}, {len(coords)} rows): fused vs oracle {e_fused:.2e}, two launches vs oracle {e_two:.2e}, fused vs two {d:.2e}")
    assert e_fused < REL and e_two < REL and d < 2e-6
    assert e_fused <= 2 * e_two + 1e-6


@pytest.mark.parametrize("ks", [5, 3])
def test_staged_first_convolution_matches_the_probing_kernel_and_the_oracle(ks):
    """conv1_bf_kernel (Z-ordered maps: the block feature vectors of a 256-parent tile's neighbourhood staged in LDS, the fine rows
    grouped by parity class, a K = 27 product over level-1 blocks) against conv1_mfma_kernel (octree probing per fine row, products
    per window position) - the same products in another order, so the features agree to fp32 rounding - and against the oracle; two clouds in a batch, non-unit features with planted zeros, both
    window sizes."""
    import eyoc_amd
    from eyoc_amd import synthetic as syn
    from oracle import resunet as orr
    L, lib = _lib()
    p = syn.make_pair(9, beams=32, azimuths=1000, band=None)
    coords = syn.batch_coords([p["coords0"], p["coords1"]])
    rng = np.random.default_rng(ks)
    feats = rng.uniform(0.25, 2.0, size=(len(coords), 1)).astype(np.float32)
    feats[rng.random(len(coords)) < 0.05] = 0.0
    sd = syn.make_weights(seed=11, in_channels=1, conv1_kernel_size=ks)
    m = eyoc_amd.load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, conv1_kernel_size=ks, normalize_feature=True)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m = m.cuda().eval()
    m.spconv_math = "split16"
    want = orr.resunet_forward(sd, coords, feats, conv1_kernel_size=ks).numpy()
    prev_order = L.knob("eyoc_maps_internal_order", 1) - 2                   # Z-ordered maps whatever the size
    prev_c1 = L.knob("eyoc_spconv_select_conv1_kernel", -1)
    outs = {}
    try:
        for staged in (0, 1):
            L.knob("eyoc_spconv_select_conv1_kernel", staged)
            outs[staged] = _forward(m, coords, feats)
            assert m.last_spconv_math == "split16"
    finally:
        L.knob("eyoc_spconv_select_conv1_kernel", prev_c1)
        L.knob("eyoc_maps_internal_order", prev_order)
    e1, e0 = rel_err(outs[1], want), rel_err(outs[0], want)
    d = rel_err(outs[1], outs[0])
    print(f"first convolution {ks}^3, {len(coords)} rows: staged vs oracle {e1:.2e}, probing vs oracle {e0:.2e}, staged vs probing {d:.2e}")
    assert e1 < REL and e0 < REL and d < 2e-6


def test_small_input_split_over_input_blocks_matches_the_unsplit_forward():
    """A single ~30k-voxel cloud through the Z-ordered split16 forward: the deep levels' stride-1 layers are a few dozen workgroups,
    so eyoc_model_forward splits every tile's 32-channel input blocks over several workgroups (eyoc_spconv_st_ksplit; partial sums
    added in share order by a second launch).  Same features as the unsplit kernel to fp32 rounding, both at the oracle's bar, and
    bit-identical from run to run."""
    from eyoc_amd import synthetic as syn
    from oracle import resunet as orr
    from test_gpu_round2 import _model
    L, lib = _lib()
    p = syn.make_pair(1)
    coords = syn.batch_coords([p["coords0"]])
    model, sd = _model()
    model.spconv_math = "split16"
    want = orr.resunet_forward(sd, coords, p["feats0"]).numpy()
    prev = L.knob("eyoc_spconv_st_ksplit", -1)
    prev_k = L.knob("eyoc_spconv_select_split16_kernel", 1)          # the default selection (the file's fixture forces the gathering kernels)
    outs = {}
    try:
        for on in (0, 1):
            L.knob("eyoc_spconv_st_ksplit", on)
            outs[on] = [_forward(model, coords, p["feats0"]) for _ in range(3 if on else 1)]
    finally:
        L.knob("eyoc_spconv_st_ksplit", prev)
        L.knob("eyoc_spconv_select_split16_kernel", prev_k)
    assert prev == 1
    for o in outs[1][1:]:
        np.testing.assert_array_equal(o, outs[1][0])
    e0, e1, d = rel_err(outs[0][0], want), rel_err(outs[1][0], want), rel_err(outs[1][0], outs[0][0])
    print(f"split over input blocks: vs oracle {e1:.2e} (unsplit {e0:.2e}), split vs unsplit {d:.2e}")
    assert e0 < REL and e1 < REL and 0 < d < 2e-6
