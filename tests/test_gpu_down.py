"""Strided convolutions (model/resunet.py:44-77: conv2 / conv3 / conv4, kernel 3, stride 2) through the staged kernel on 128-row output
tiles (round 6; spconv_st.hip ``launch_spconv_st128``, records from ``k_local_rulebook<.., 2>`` on the strided tables) against the
gathering kernel they replace and, through the whole forward, against the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def batches_take_the_batch_kernels():
    """class-major transposed records, lazy tables and the staged strided kernel start at 2^17 rows in production: lowered so that a
    two-cloud batch (62 k rows) runs them"""
    from eyoc_amd import _lib
    prev = _lib.knob("eyoc_spconv_upc_min_rows", 8192)
    yield
    _lib.knob("eyoc_spconv_upc_min_rows", prev)
    _lib.knob("eyoc_spconv_select_down_kernel", 1)


def _forward(model, coords, feats):
    import eyoc_amd
    x = eyoc_amd.SparseTensor(torch.from_numpy(feats).cuda(), coordinates=torch.from_numpy(coords).cuda())
    return model(x).F.clone()


@pytest.mark.parametrize("seed", [2, 7])
def test_staged_strided_layers_equal_the_gathering_kernel_and_the_oracle(batches_take_the_batch_kernels, seed):
    from eyoc_amd import _lib, synthetic as syn
    from test_gpu_round2 import _model
    p = syn.make_pair(seed)
    coords = syn.batch_coords([p["coords0"], p["coords1"]])
    feats = np.random.default_rng(seed).uniform(0.5, 1.5, size=(len(coords), 1)).astype(np.float32)
    model, sd = _model()
    assert _lib.knob("eyoc_spconv_select_down_kernel", -7) == 1          # the default
    f_staged = _forward(model, coords, feats)
    _lib.knob("eyoc_spconv_select_down_kernel", 0)
    f_gather = _forward(model, coords, feats)
    _lib.knob("eyoc_spconv_select_down_kernel", 1)
    assert not torch.equal(f_staged, f_gather), "both runs took the same kernel - the switch did nothing"
    # same products, another summation order (32-channel blocks outside the offsets instead of inside): fp32 rounding only
    assert float((f_staged - f_gather).abs().max()) < 2e-5
    assert torch.equal(f_staged, _forward(model, coords, feats))        # reproducible
    from oracle import resunet as orr
    want = np.asarray(orr.resunet_forward(sd, coords, feats))
    err = np.abs(f_staged.cpu().numpy() - want).max()
    assert err <= 1e-4 * np.abs(want).max(), err


def test_a_cloud_in_no_spatial_order_falls_back(batches_take_the_batch_kernels):
    """Rows in random positions: a 128-row coarse tile then reads more distinct fine rows than two stage passes hold (or its hash
    fills up) - the table keeps the gathering kernel, and the forward still matches the oracle."""
    from test_gpu_round2 import _model
    from oracle import resunet as orr
    rng = np.random.default_rng(0)
    c = np.unique(rng.integers(0, 30, size=(24000, 3)), axis=0).astype(np.int32)       # a dense 30^3 box: every coarse row has ~27 fine neighbours
    from eyoc_amd import synthetic as syn
    coords = syn.batch_coords([c])
    feats = rng.uniform(0.5, 1.5, size=(len(coords), 1)).astype(np.float32)
    model, sd = _model()
    f = _forward(model, coords, feats).cpu().numpy()
    want = np.asarray(orr.resunet_forward(sd, coords, feats))
    assert np.abs(f - want).max() <= 1e-4 * np.abs(want).max()


@pytest.mark.parametrize("case", ["tiny", "random", "one_cloud_of_a_pair"])
def test_batch_kernels_on_small_and_odd_inputs(case):
    """Robustness of the round-6 paths away from their production sizes: Z-order, class-major transposed records, lazy tables, 128-row
    strided tiles, the wide 128-row stride-1 tiles and the tail in the last layer's epilogue FORCED on inputs of 6, ~2000 and ~30 000
    rows (a ragged only tile, levels with fewer rows than a tile, a coarsest level of a handful of rows) - tables and forward against
    the oracle."""
    import eyoc_amd
    from eyoc_amd import _lib, synthetic as syn
    from oracle import coords as oc, resunet as orr
    from test_gpu_round2 import _model
    if case == "tiny":
        coords = np.array([[0, -1, -1, -1], [0, -2, 0, 1], [0, 0, 0, 0], [1, 0, 0, 0], [1, 1, 0, 0], [0, -9, 7, -8]], np.int32)
    elif case == "random":
        rng = np.random.default_rng(4)
        c = np.unique(rng.integers(-12, 12, size=(2500, 3)), axis=0).astype(np.int32)
        rng.shuffle(c)
        coords = syn.batch_coords([c[:1200], c[1200:]])
    else:
        coords = syn.batch_coords([syn.make_pair(11)["coords1"]])
    feats = np.random.default_rng(1).uniform(0.5, 1.5, size=(len(coords), 1)).astype(np.float32)
    model, sd = _model()
    knobs = [("eyoc_spconv_upc_min_rows", 0), ("eyoc_maps_internal_order", 1), ("eyoc_spconv_st_split_below", 0)]
    prev = [(k, _lib.knob(k, v)) for k, v in knobs]
    model.spconv_math = "split16"
    try:
        x = eyoc_amd.SparseTensor(torch.from_numpy(feats).cuda(), coordinates=torch.from_numpy(coords).cuda())
        f = model(x).F.cpu().numpy()
        assert model.last_spconv_math == "split16"
        cm = x.coordinate_manager
        perm = cm.row_order().cpu().numpy()
        want_maps = oc.build_maps(coords[perm])
        for l in range(4):
            np.testing.assert_array_equal(cm.table(_lib.MAP_S1, l, internal=True).cpu().numpy(), want_maps["s1"][l])
            if l < 3:
                np.testing.assert_array_equal(cm.table(_lib.MAP_UP, l, internal=True).cpu().numpy(), want_maps["up"][l])
                np.testing.assert_array_equal(cm.table(_lib.MAP_DOWN, l, internal=True).cpu().numpy(), want_maps["down"][l])
    finally:
        model.spconv_math = "auto"
        for k, v in prev:
            _lib.knob(k, v if k != "eyoc_maps_internal_order" else v - 2)
    want = np.asarray(orr.resunet_forward(sd, coords, feats))
    ok = np.isfinite(want).all(axis=1)                                   # (an isolated voxel can normalise 0 / 0 in both)
    assert np.abs(f[ok] - want[ok]).max() <= 1e-4 * np.abs(want[ok]).max()


def test_workgroup_to_tile_mappings_and_tile_shapes_give_the_same_bits(batches_take_the_batch_kernels):
    """Switches that only change WHICH workgroup computes a tile or how a tile is cut over workgroups - the channel groups of a tile on
    one XCD (eyoc_spconv_st_ksplit 2 / 3), 64- or 128-channel workgroups on 128-row tiles (eyoc_spconv_select_down_kernel 2 / 3), the
    lazy tables - must not change a bit of the forward: every output element is the same products summed in the same order.  The
    256-channel stride-1 layers on 128- or 256-row tiles (4 / 5) are the exception that proves it: another tile shape means another
    split of a two-pass tile's input rows over its passes, i.e. another summation order for those tiles - fp32 rounding, nothing more."""
    from eyoc_amd import _lib, synthetic as syn
    from test_gpu_round2 import _model
    p = syn.make_pair(13)
    coords = syn.batch_coords([p["coords0"], p["coords1"]])
    feats = np.random.default_rng(13).uniform(0.5, 1.5, size=(len(coords), 1)).astype(np.float32)
    model, _sd = _model()
    base = _forward(model, coords, feats)
    try:
        for knob, off, on in (("eyoc_spconv_st_ksplit", 2, 3), ("eyoc_spconv_select_down_kernel", 2, 3), ("eyoc_spconv_select_down_kernel", 4, 5),
                              ("eyoc_maps_lazy_tables", 0, 1)):
            _lib.knob(knob, off)
            alt = _forward(model, coords, feats)
            _lib.knob(knob, on)
            if (off, on) == (4, 5):
                d = float((alt - base).abs().max())
                assert 0.0 < d < 2e-6, d
            else:
                assert torch.equal(alt, base), (knob, off)
    finally:
        _lib.knob("eyoc_spconv_st_ksplit", 3)
        _lib.knob("eyoc_spconv_select_down_kernel", 3)
        _lib.knob("eyoc_spconv_select_down_kernel", 5)
        _lib.knob("eyoc_maps_lazy_tables", 1)
