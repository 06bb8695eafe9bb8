"""GPU parity: sparse-convolution layers and the whole ResUNet forward through the C ABI vs the CPU
oracle.  Floating point: the bar is 1e-4 relative (BASELINE.json north_star); the tests use
max |a - b| <= 1e-4 * max|b| per tensor and a tighter row-wise check on the unit-norm features."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REL = 1e-4


@pytest.fixture(params=[0, 1], ids=["tiled", "wave"], autouse=True)
def spconv_kernel(request):
    """Every test runs against both decompositions of the sparse convolution (spconv.hip / spconv_wave.hip);
    production picks between them by problem size."""
    from eyoc_amd import _lib
    lib = _lib.load()
    prev = _lib.knob("eyoc_spconv_select_kernel", request.param)
    prev_rows = _lib.knob("eyoc_maps_order_min_rows", 0)   # build the tiling orders for the small test clouds as well
    yield request.param
    _lib.knob("eyoc_spconv_select_kernel", prev)
    _lib.knob("eyoc_maps_order_min_rows", prev_rows)


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def run_layer(nbr, x, W, bias=None, scale=None, res=None, relu=False, ld_pad=(0, 0, 0), n_out=None):
    """One eyoc_spconv call; ld_pad adds unused leading/trailing columns to in/out/res to exercise
    the leading-dimension and column-offset arguments (the concat-buffer case)."""
    from eyoc_amd import _lib
    lib = _lib.load()
    K, cin, cout = W.shape
    packed = np.zeros(K * cin * cout, np.float32)
    sc = None if scale is None else np.ascontiguousarray(scale, np.float32)
    assert lib.eyoc_spconv_pack_weights(np.ascontiguousarray(W).ctypes.data, None if sc is None else sc.ctypes.data, K,
                                        cin, cout, packed.ctypes.data) == 0
    dev = torch.device("cuda")
    n_out = n_out if n_out is not None else (nbr.shape[1] if nbr is not None else x.shape[0])

    def padded(a, pad):
        if a is None:
            return None, 0, None
        full = torch.full((a.shape[0], a.shape[1] + 2 * pad), 777.0, device=dev)
        full[:, pad:pad + a.shape[1]] = torch.from_numpy(a).to(dev)
        return full, full.shape[1], full[:, pad:]

    xin, ld_in, xin_v = padded(x, ld_pad[0])
    out_full = torch.full((n_out, cout + 2 * ld_pad[1]), -555.0, device=dev)
    out_v = out_full[:, ld_pad[1]:]
    rfull, ld_res, r_v = padded(res, ld_pad[2])
    wd = torch.from_numpy(packed).to(dev)
    bd = None if bias is None else torch.from_numpy(np.ascontiguousarray(bias, np.float32)).to(dev)
    nd = None if nbr is None else torch.from_numpy(np.ascontiguousarray(nbr, np.int32)).to(dev)
    _lib.check(lib.eyoc_spconv(_lib.ctx(), _lib.ptr(nd), K, n_out, C.c_void_p(xin_v.data_ptr()), ld_in, cin, _lib.ptr(wd),
                               cout, _lib.ptr(bd), None if r_v is None else C.c_void_p(r_v.data_ptr()), ld_res,
                               1 if relu else 0, C.c_void_p(out_v.data_ptr()), out_full.shape[1], _lib.stream_ptr()),
               "eyoc_spconv")
    torch.cuda.synchronize()
    o = out_full.cpu().numpy()
    if ld_pad[1]:
        assert (o[:, :ld_pad[1]] == -555.0).all() and (o[:, -ld_pad[1]:] == -555.0).all(), "wrote outside its columns"
    return o[:, ld_pad[1]:ld_pad[1] + cout]


def oracle_layer(nbr, x, W, bias=None, scale=None, res=None, relu=False):
    from oracle import resunet as orr
    if nbr is None:
        nbr = np.arange(x.shape[0], dtype=np.int32)[None]
    out = orr.sparse_conv(torch.from_numpy(x), nbr, torch.from_numpy(W)).numpy()
    if scale is not None:
        out = out * scale[None]
    if bias is not None:
        out = out + bias[None]
    if res is not None:
        out = out + res
    return np.maximum(out, 0) if relu else out


def small_maps(seed=0, n=2500):
    from oracle import coords as oc
    from eyoc_amd import synthetic as syn
    rng = np.random.default_rng(seed)
    c = np.unique(rng.integers(-14, 14, size=(n, 3)) // np.array([1, 1, 3]), axis=0)
    rng.shuffle(c)
    return oc.build_maps(syn.batch_coords([c.astype(np.int32)]))


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 64), (32, 64), (128, 128), (256, 256), (256, 64), (128, 256), (96, 64)])
def test_stride1_layer_shapes(cin, cout):
    maps = small_maps()
    nbr = maps["s1"][0]
    rng = np.random.default_rng(cin * 1000 + cout)
    x = rng.normal(size=(nbr.shape[1], cin)).astype(np.float32)
    W = (rng.normal(size=(27, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    got = run_layer(nbr, x, W)
    assert rel_err(got, oracle_layer(nbr, x, W)) < REL


def test_epilogue_bias_residual_relu_and_column_offsets():
    maps = small_maps(1)
    nbr = maps["s1"][0]
    n = nbr.shape[1]
    rng = np.random.default_rng(5)
    x = rng.normal(size=(n, 64)).astype(np.float32)
    W = (rng.normal(size=(27, 64, 64)) / 24).astype(np.float32)
    b = rng.normal(size=64).astype(np.float32)
    s = rng.uniform(0.5, 1.5, 64).astype(np.float32)
    r = rng.normal(size=(n, 64)).astype(np.float32)
    got = run_layer(nbr, x, W, bias=b, scale=s, res=r, relu=True, ld_pad=(32, 64, 4))
    want = oracle_layer(nbr, x, W, bias=b, scale=s, res=r, relu=True)
    assert rel_err(got, want) < REL
    assert (got >= 0).all() and (got == 0).any()


def test_strided_transposed_and_identity_maps():
    maps = small_maps(2)
    rng = np.random.default_rng(6)
    n1, n2 = len(maps["cm"][0]), len(maps["cm"][1])
    x1 = rng.normal(size=(n1, 32)).astype(np.float32)
    Wd = (rng.normal(size=(27, 32, 64)) / 16).astype(np.float32)
    down = run_layer(maps["down"][0], x1, Wd)
    assert down.shape == (n2, 64) and rel_err(down, oracle_layer(maps["down"][0], x1, Wd)) < REL
    Wu = (rng.normal(size=(27, 64, 32)) / 16).astype(np.float32)
    up = run_layer(maps["up"][0], down, Wu)
    assert up.shape == (n1, 32) and rel_err(up, oracle_layer(maps["up"][0], down, Wu)) < REL
    W1 = (rng.normal(size=(1, 96, 64)) / 10).astype(np.float32)
    x96 = rng.normal(size=(n1, 96)).astype(np.float32)
    assert rel_err(run_layer(None, x96, W1, relu=True), oracle_layer(None, x96, W1, relu=True)) < REL


@pytest.mark.parametrize("n", [1, 15, 16, 17, 63, 64, 65, 200, 5000])
def test_ragged_row_counts(n):
    """Row counts around the 16-pair chunk and the 32/64/128-row tile boundaries."""
    from oracle import coords as oc
    rng = np.random.default_rng(n)
    c = np.unique(rng.integers(-6, 6, size=(4 * n + 8, 3)), axis=0)
    rng.shuffle(c)
    c = c[:n]
    coords = np.concatenate([np.zeros((len(c), 1), np.int64), c], 1)
    cm = oc.CoordMap(coords, 1)
    nbr = oc.kernel_map(cm, cm, 3)
    x = rng.normal(size=(len(c), 32)).astype(np.float32)
    W = (rng.normal(size=(27, 32, 32)) / 10).astype(np.float32)
    assert rel_err(run_layer(nbr, x, W), oracle_layer(nbr, x, W)) < REL


def test_layer_is_deterministic():
    maps = small_maps(3)
    nbr = maps["s1"][0]
    rng = np.random.default_rng(7)
    x = rng.normal(size=(nbr.shape[1], 64)).astype(np.float32)
    W = (rng.normal(size=(27, 64, 64)) / 24).astype(np.float32)
    a, b = run_layer(nbr, x, W), run_layer(nbr, x, W)
    np.testing.assert_array_equal(a, b)


def test_bad_shapes_fail_loudly():
    import eyoc_amd
    maps = small_maps(4, 300)
    nbr = maps["s1"][0]
    x = np.zeros((nbr.shape[1], 48), np.float32)
    with pytest.raises(AssertionError):
        run_layer(nbr, x, np.zeros((27, 48, 64), np.float32))     # pack rejects C_in % 32 != 0


# ---------------------------------------------------------------------------------------- whole network
def make_model(sd, name="ResUNetBN2C", **kw):
    import eyoc_amd
    Model = eyoc_amd.load_model(name)
    model = Model(kw.pop("in_channels", 1), kw.pop("out_channels", 32), bn_momentum=0.05,
                  conv1_kernel_size=kw.pop("conv1_kernel_size", 5), normalize_feature=kw.pop("normalize_feature", True))
    missing = model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    assert not missing.missing_keys and not missing.unexpected_keys
    return model.cuda().eval()


def check_forward(coords, feats, sd, **kw):
    import eyoc_amd
    from oracle import resunet as orr
    model = make_model(sd, **kw)
    x = eyoc_amd.SparseTensor(torch.from_numpy(feats).cuda(), coordinates=torch.from_numpy(coords).cuda())
    with torch.no_grad():
        out = model(x)
    assert isinstance(out, eyoc_amd.SparseTensor) and out.coordinate_manager is x.coordinate_manager
    got = out.F.cpu().numpy()
    want = orr.resunet_forward(sd, coords, feats, normalize_feature=kw.get("normalize_feature", True),
                               conv1_kernel_size=kw.get("conv1_kernel_size", 5)).numpy()
    assert got.shape == want.shape
    assert rel_err(got, want) < REL, rel_err(got, want)
    return got, want


def test_resunet_forward_small_batched_cloud():
    from eyoc_amd import synthetic as syn
    rng = np.random.default_rng(11)
    c = np.unique(rng.integers(-12, 12, size=(3000, 3)) // np.array([1, 1, 2]), axis=0)
    rng.shuffle(c)
    coords = syn.batch_coords([c[:900].astype(np.int32), c[900:].astype(np.int32)])
    sd = syn.make_weights()
    got, want = check_forward(coords, np.ones((len(coords), 1), np.float32), sd)
    np.testing.assert_allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    # row order contract: output row i belongs to input coordinate i (scripts/test_kitti.py:28-42)
    perm = rng.permutation(len(coords))
    got_p, _ = check_forward(coords[perm], np.ones((len(coords), 1), np.float32), sd)
    assert rel_err(got_p, got[perm]) < 1e-5


def test_resunet_forward_synthetic_kitti_cloud():
    from eyoc_amd import synthetic as syn
    p = syn.make_pair(1)
    coords = syn.batch_coords([p["coords0"]])
    got, want = check_forward(coords, p["feats0"], syn.make_weights())
    assert 26000 < len(got) < 35000
    cos = (got * want).sum(1)
    assert cos.min() > 1 - 1e-6


def test_resunet_unnormalised_output_and_other_channel_tables():
    from eyoc_amd import synthetic as syn
    rng = np.random.default_rng(12)
    c = np.unique(rng.integers(-10, 10, size=(1500, 3)), axis=0).astype(np.int32)
    coords = syn.batch_coords([c])
    feats = np.ones((len(coords), 1), np.float32)
    check_forward(coords, feats, syn.make_weights(), normalize_feature=False)
    # ResUNetBN2B channel table, 3 input channels, 3^3 first conv
    sd = syn.make_weights(seed=5, in_channels=3, conv1_kernel_size=3, tr_channels=(None, 64, 64, 64, 64))
    f3 = rng.normal(size=(len(coords), 3)).astype(np.float32)
    check_forward(coords, f3, sd, name="ResUNetBN2B", in_channels=3, conv1_kernel_size=3)


def test_model_api_surface():
    import eyoc_amd
    assert eyoc_amd.load_model("NoSuchNet") is None
    Model = eyoc_amd.load_model("ResUNetBN2C")
    m = Model(1, 32, bn_momentum=0.05, conv1_kernel_size=5, normalize_feature=True)
    keys = set(m.state_dict().keys())
    for k in ("conv1.kernel", "norm1.bn.running_mean", "block1.conv2.kernel", "block4_tr.norm2.bn.weight",
              "conv1_tr.kernel", "final.kernel", "final.bias", "norm2_tr.bn.num_batches_tracked"):
        assert k in keys, k
    assert tuple(m.state_dict()["conv1.kernel"].shape) == (125, 1, 32)
    assert tuple(m.state_dict()["conv3_tr.kernel"].shape) == (27, 256, 64)
    assert tuple(m.state_dict()["conv1_tr.kernel"].shape) == (96, 64)
    assert tuple(m.state_dict()["final.bias"].shape) == (1, 32)
    x = eyoc_amd.SparseTensor(torch.ones(4, 1).cuda(), coordinates=torch.tensor([[0, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0],
                                                                               [0, 5, 5, 5]], dtype=torch.int32).cuda())
    tr = m.cuda().train()(x)                              # training mode: batch statistics + autograd (tests/test_gpu_train.py)
    assert tr.F.shape == (4, 32) and tr.F.requires_grad
    out = m.cuda().eval()(x)
    assert out.F.shape == (4, 32) and len(out) == 4 and out.C.shape == (4, 4)
    cs, fs = out.decomposed_coordinates_and_features
    assert len(cs) == 1 and fs[0].shape == (4, 32)


def test_first_conv_kernel_size_7_hash_probing_fallback():
    """7^3 does not fit the octree walk (3^3 / 5^3 only): the first convolution probes the level-0 hash table, which
    the maps build on demand (FCGF's 3DMatch setting uses conv1_kernel_size = 7)."""
    from eyoc_amd import synthetic as syn
    rng = np.random.default_rng(13)
    c = np.unique(rng.integers(-9, 9, size=(1200, 3)), axis=0).astype(np.int32)
    coords = syn.batch_coords([c[:500], c[500:]])
    sd = syn.make_weights(seed=7, conv1_kernel_size=7)
    check_forward(coords, np.ones((len(coords), 1), np.float32), sd, conv1_kernel_size=7)
