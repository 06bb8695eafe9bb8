"""GPU parity: coordinate maps and rulebooks built by the HIP library vs the oracle (integer work:
bit-exact, including row order)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def random_cloud(seed, n, extent=40, batch=1):
    rng = np.random.default_rng(seed)
    out = []
    for b in range(batch):
        c = np.unique(rng.integers(-extent, extent, size=(n, 3)) // np.array([1, 1, 4]), axis=0)
        rng.shuffle(c)
        out.append(c.astype(np.int32))
    from eyoc_amd import synthetic as syn
    return syn.batch_coords(out)


@pytest.fixture(autouse=True)
def order_small_levels():
    """The tiling orders are only built for large levels in production; build them for the small test clouds too."""
    from eyoc_amd import _lib
    lib = _lib.load()
    prev = _lib.knob("eyoc_maps_order_min_rows", 0)
    yield
    _lib.knob("eyoc_maps_order_min_rows", prev)


def check_up_order(order, up, window_shift=None):
    """The tiling order of a transposed convolution: a permutation of the rows in which every pattern of
    occupied offsets forms ONE run, rows ascending inside a run (stable).  Z-ordered maps order inside windows of
    2^window_shift consecutive rows (windows ascending, one run per pattern and window)."""
    n = up.shape[1]
    np.testing.assert_array_equal(np.sort(order), np.arange(n))
    window = np.zeros(n, np.int64) if window_shift is None else (order >> window_shift).astype(np.int64)
    assert (np.diff(window) >= 0).all(), "windows are not in ascending order"
    pattern = ((up >= 0) * (1 << np.arange(27, dtype=np.int64))[:, None]).sum(0)[order] + (window << 27)
    starts = np.flatnonzero(np.r_[True, pattern[1:] != pattern[:-1]])
    assert len(starts) == len(np.unique(pattern)), "a pattern is split over several runs"
    same = pattern[1:] == pattern[:-1]
    assert (np.diff(order)[same] > 0).all(), "rows of one pattern are not in ascending order"


def check_against_oracle(coords):
    import eyoc_amd
    from eyoc_amd import _lib
    from oracle import coords as oc
    cm = eyoc_amd.CoordinateManager(torch.from_numpy(coords).cuda())
    maps = oc.build_maps(coords, conv1_kernel_size=5)
    info = cm.info(conv1_kernel_size=5)
    st = oc.map_stats(maps)
    assert info["rows"] == st["rows"]
    assert info["pairs_s1"] == st["pairs_s1"] and info["pairs_down"] == st["pairs_down"]
    assert info["pairs_up"] == st["pairs_up"] and info["pairs_conv1"] == st["pairs_k5"]
    for l in range(4):
        np.testing.assert_array_equal(cm.level_coordinates(l).cpu().numpy(), maps["cm"][l].coords)
        np.testing.assert_array_equal(cm.table(_lib.MAP_S1, l).cpu().numpy(), maps["s1"][l])
        if l < 3:
            np.testing.assert_array_equal(cm.table(_lib.MAP_DOWN, l).cpu().numpy(), maps["down"][l])
            np.testing.assert_array_equal(cm.table(_lib.MAP_UP, l).cpu().numpy(), maps["up"][l])
            check_up_order(cm.up_order(l).cpu().numpy(), maps["up"][l])
    return info


def test_maps_small_random_cloud():
    check_against_oracle(random_cloud(0, 3000))


def test_maps_batched_clouds_do_not_mix():
    info = check_against_oracle(random_cloud(1, 2500, batch=3))
    assert info["rows"][0] > 5000


def test_maps_negative_coordinates_and_tiny_inputs():
    coords = np.array([[0, -1, -1, -1], [0, -2, 0, 1], [0, 0, 0, 0], [1, 0, 0, 0], [1, 1, 0, 0], [0, -9, 7, -8]], np.int32)
    check_against_oracle(coords)
    check_against_oracle(np.array([[0, 5, 5, 5]], np.int32))


def test_maps_synthetic_kitti_cloud():
    from eyoc_amd import synthetic as syn
    p = syn.make_pair(0)
    info = check_against_oracle(syn.batch_coords([p["coords0"], p["coords1"]]))
    assert 57000 <= info["rows"][0] <= 66000


@pytest.fixture
def zorder_rows():
    from eyoc_amd import _lib
    lib = _lib.load()
    prev = _lib.knob("eyoc_maps_internal_order", 1) - 2
    prev_w = _lib.knob("eyoc_maps_order_window_shift", 12)   # several windows in a 62k-row cloud
    # the windowed tiling order of the transposed tables is only built when the staged transposed kernel (which sorts inside its
    # tiles instead) is off: switch it off for these builds so that the order is still checked
    prev_up = _lib.knob("eyoc_spconv_select_up_kernel", 0)
    yield
    _lib.knob("eyoc_maps_internal_order", prev)
    _lib.knob("eyoc_maps_order_window_shift", prev_w)
    _lib.knob("eyoc_spconv_select_up_kernel", prev_up)


@pytest.mark.parametrize("case", ["random3", "kitti", "tiny"])
def test_maps_in_z_order_equal_the_oracle_maps_of_the_z_ordered_cloud(zorder_rows, case):
    """Large batches keep their rows in Z-order internally (eyoc_maps_row_order); forced here for small clouds.  The
    permutation is the stable Morton order of (batch, x, y, z) and every level, table and tiling order is bit-exactly
    what the oracle builds for the cloud permuted that way."""
    import eyoc_amd
    from eyoc_amd import synthetic as syn
    from test_gpu_split16 import morton_order
    if case == "random3":
        coords = random_cloud(5, 2500, batch=3)
    elif case == "kitti":
        p = syn.make_pair(3)
        coords = syn.batch_coords([p["coords0"], p["coords1"]])
    else:
        coords = np.array([[0, -1, -1, -1], [0, -2, 0, 1], [0, 0, 0, 0], [1, 0, 0, 0], [1, 1, 0, 0], [0, -9, 7, -8]], np.int32)
    cm = eyoc_amd.CoordinateManager(torch.from_numpy(coords).cuda())
    perm = cm.row_order().cpu().numpy()
    np.testing.assert_array_equal(perm, morton_order(coords))
    np.testing.assert_array_equal(cm.level_coordinates(0, internal=True).cpu().numpy(), coords[perm])
    from oracle import coords as oc
    from eyoc_amd import _lib
    maps = oc.build_maps(coords[perm], conv1_kernel_size=5)
    info = cm.info(conv1_kernel_size=5)
    assert info["rows"] == oc.map_stats(maps)["rows"] and info["pairs_conv1"] == oc.map_stats(maps)["pairs_k5"]
    for l in range(4):
        np.testing.assert_array_equal(cm.level_coordinates(l, internal=True).cpu().numpy(), maps["cm"][l].coords)
        np.testing.assert_array_equal(cm.table(_lib.MAP_S1, l, internal=True).cpu().numpy(), maps["s1"][l])
        if l < 3:
            np.testing.assert_array_equal(cm.table(_lib.MAP_DOWN, l, internal=True).cpu().numpy(), maps["down"][l])
            np.testing.assert_array_equal(cm.table(_lib.MAP_UP, l, internal=True).cpu().numpy(), maps["up"][l])
            check_up_order(cm.up_order(l, internal=True).cpu().numpy(), maps["up"][l], window_shift=12)


def test_z_order_sort_with_a_speculated_key_width_falls_back_to_the_full_width(zorder_rows):
    """The Morton key is as wide as the previous build of the context needed (fewer radix passes); a batch that does not fit -
    larger coordinates, a larger batch index - is sorted again with the full width: same permutation as the stable Morton
    order either way, in whatever order the builds come."""
    import eyoc_amd
    from test_gpu_split16 import morton_order
    rng = np.random.default_rng(11)
    def cloud(extent, batch):
        c = np.unique(rng.integers(-extent, extent, size=(3000, 3)), axis=0).astype(np.int32)
        rng.shuffle(c)
        return np.concatenate([np.full((len(c), 1), batch, np.int32), c], 1)
    for extent, batch in ((20, 0), (4000, 0), (20, 0), (300, 700), (60000, 3), (5, 1)):
        coords = cloud(extent, batch)
        cm = eyoc_amd.CoordinateManager(torch.from_numpy(coords).cuda())
        np.testing.assert_array_equal(cm.row_order().cpu().numpy(), morton_order(coords), err_msg=f"extent {extent}, batch {batch}")


def test_maps_keep_the_callers_order_for_small_clouds():
    import eyoc_amd
    cm = eyoc_amd.CoordinateManager(torch.from_numpy(random_cloud(6, 500)).cuda())
    assert cm.row_order() is None


def test_maps_reject_duplicates_and_out_of_range():
    import eyoc_amd
    dup = torch.tensor([[0, 1, 2, 3], [0, 4, 5, 6], [0, 1, 2, 3]], dtype=torch.int32).cuda()
    with pytest.raises(eyoc_amd.EyocError, match="duplicate"):
        eyoc_amd.CoordinateManager(dup).maps()
    far = torch.tensor([[0, 1 << 17, 0, 0]], dtype=torch.int32).cuda()
    with pytest.raises(eyoc_amd.EyocError, match="range"):
        eyoc_amd.CoordinateManager(far).maps()
    with pytest.raises(eyoc_amd.EyocError):
        eyoc_amd.CoordinateManager(torch.zeros((0, 4), dtype=torch.int32).cuda()).maps()


def test_z_ordered_maps_reject_duplicates_and_out_of_range(zorder_rows):
    """The hash-free level construction of Z-ordered maps (adjacent rows of the sorted list) keeps the validation."""
    import eyoc_amd
    rng = np.random.default_rng(3)
    base = np.unique(rng.integers(-30, 30, size=(4000, 3)), axis=0).astype(np.int32)
    coords = np.concatenate([np.zeros((len(base), 1), np.int32), base], 1)
    dup = np.concatenate([coords, coords[100:101]])                       # one duplicate, far apart in the caller's order
    with pytest.raises(eyoc_amd.EyocError, match="duplicate"):
        eyoc_amd.CoordinateManager(torch.from_numpy(dup).cuda()).maps()
    far = coords.copy()
    far[7, 1] = 1 << 17
    with pytest.raises(eyoc_amd.EyocError, match="range"):
        eyoc_amd.CoordinateManager(torch.from_numpy(far).cuda()).maps()
    eyoc_amd.CoordinateManager(torch.from_numpy(coords).cuda()).maps()    # and the clean cloud builds


def test_accessors_answer_in_the_callers_rows_after_a_z_ordered_forward():
    """A forward on >= 8192 rows builds Z-ordered maps.  The accessors (what the autograd layer functions index the caller's
    feature rows with) still answer in the CALLER's rows - from a second map set built on first use - and equal the oracle's
    tables of the cloud as given; ``internal=True`` hands out the forward's own rows."""
    import eyoc_amd
    from eyoc_amd import _lib, synthetic as syn
    from oracle import coords as oc
    from test_gpu_round2 import _model
    p = syn.make_pair(1)
    coords = syn.batch_coords([p["coords0"]])
    model, _sd = _model()
    x = eyoc_amd.SparseTensor(torch.from_numpy(p["feats0"]).cuda(), coordinates=torch.from_numpy(coords).cuda())
    f1 = model(x).F
    cm = x.coordinate_manager
    perm = cm.row_order()
    assert perm is not None                                              # the forward's maps are Z-ordered
    want = oc.build_maps(coords)
    np.testing.assert_array_equal(cm.table(_lib.MAP_S1, 0).cpu().numpy(), want["s1"][0])
    np.testing.assert_array_equal(cm.level_coordinates(1).cpu().numpy(), want["cm"][1].coords)
    np.testing.assert_array_equal(cm.table(_lib.MAP_DOWN, 0).cpu().numpy(), want["down"][0])
    assert cm.row_order() is not None                                    # ... and the forward's maps were left alone
    inner = cm.level_coordinates(0, internal=True).cpu().numpy()
    np.testing.assert_array_equal(inner, coords[perm.cpu().numpy()])
    # the other way round: accessors first -> the caller's order, and the forward then runs on those maps
    x2 = eyoc_amd.SparseTensor(torch.from_numpy(p["feats0"]).cuda(), coordinates=torch.from_numpy(coords).cuda())
    t = x2.coordinate_manager.table(0, 0)
    assert x2.coordinate_manager.row_order() is None and t.shape == (27, len(coords))
    f2 = model(x2).F
    assert float((f1 - f2).abs().max()) < 1e-5


def test_kernel_selection_switches_belong_to_the_ctx_they_are_set_on():
    """The `eyoc_*_select_*` / `eyoc_maps_*` / `eyoc_ransac_*` switches were file-scope statics until round 4 (two models in one process
    shared kernel selection).  A second ctx on the same device: what is set on it stays on it, the process's own ctx keeps its values, and a
    NULL ctx is refused (-1)."""
    import ctypes as C
    from eyoc_amd import _lib
    lib = _lib.load()
    other = C.c_void_p()
    _lib.check(lib.eyoc_create(0, C.byref(other)), "eyoc_create")
    try:
        cases = [("eyoc_spconv_select_split16_kernel", 2, 1), ("eyoc_spconv_select_up_kernel", 0, 2), ("eyoc_maps_order_window_shift", 12, 18),
                 ("eyoc_ransac_transform_store", 8, 1 << 20), ("eyoc_knn_prefilter", 0, 1), ("eyoc_spconv_st_ksplit", 0, 1)]
        for name, value, default in cases:
            mine = _lib.knob(name, -7)                                   # out of range: a query
            f = getattr(lib, name)
            assert f(other, value) == default, name                      # a fresh ctx starts from the defaults
            assert f(other, -7) == value, name
            assert _lib.knob(name, -7) == mine, name                     # ... and the process's ctx did not notice
            assert f(None, value) == -1, name
        assert lib.eyoc_maps_select_orders(other, 0, 1) == 1 and lib.eyoc_maps_select_orders(other, -1, -1) == 2
    finally:
        lib.eyoc_destroy(other)


def test_lazy_tables_change_nothing_a_caller_can_observe():
    """Round 6: a Z-ordered build with class-major transposed records derives the tile records of the finest stride-1 table and of the
    transposed tables straight from the octree links (derive.h; compact [8][n] transposed tables) and leaves those [27][n] tables
    unwritten.  The forward must be bit-identical to the eager build's, and the accessors - which fill a skipped table on first use -
    must still return the oracle's tables."""
    import eyoc_amd
    from eyoc_amd import _lib, synthetic as syn
    from oracle import coords as oc
    from test_gpu_round2 import _model
    p = syn.make_pair(2)
    coords = syn.batch_coords([p["coords0"], p["coords1"]])
    feats = np.random.default_rng(3).uniform(0.5, 1.5, size=(len(coords), 1)).astype(np.float32)
    model, _sd = _model()
    prev_min = _lib.knob("eyoc_spconv_upc_min_rows", 8192)               # class-major records (and with them lazy tables) for a 2-cloud batch
    assert _lib.knob("eyoc_maps_lazy_tables", -7) == 1                   # the default
    try:
        outs, cms = [], []
        for lazy in (1, 0):
            _lib.knob("eyoc_maps_lazy_tables", lazy)
            x = eyoc_amd.SparseTensor(torch.from_numpy(feats).cuda(), coordinates=torch.from_numpy(coords).cuda())
            outs.append(model(x).F.clone())
            cms.append(x.coordinate_manager)
        assert torch.equal(outs[0], outs[1])
        cm = cms[0]
        perm = cm.row_order().cpu().numpy()
        want = oc.build_maps(coords[perm])
        for l in range(4):
            np.testing.assert_array_equal(cm.table(_lib.MAP_S1, l, internal=True).cpu().numpy(), want["s1"][l])
            if l < 3:
                np.testing.assert_array_equal(cm.table(_lib.MAP_UP, l, internal=True).cpu().numpy(), want["up"][l])
                np.testing.assert_array_equal(cm.table(_lib.MAP_DOWN, l, internal=True).cpu().numpy(), want["down"][l])
        info = cm.info(conv1_kernel_size=5)
        st = oc.map_stats(want)
        assert info["pairs_s1"] == st["pairs_s1"] and info["pairs_up"] == st["pairs_up"] and info["pairs_down"] == st["pairs_down"]
        # a forward that needs the skipped tables (every split16 layer on the gathering kernels) fills them itself
        _lib.knob("eyoc_maps_lazy_tables", 1)
        x = eyoc_amd.SparseTensor(torch.from_numpy(feats).cuda(), coordinates=torch.from_numpy(coords).cuda())
        prev_k = _lib.knob("eyoc_spconv_select_split16_kernel", 0)
        try:
            f_gather = model(x).F
        finally:
            _lib.knob("eyoc_spconv_select_split16_kernel", prev_k)
        assert float((f_gather - outs[0]).abs().max()) < 2e-5
    finally:
        _lib.knob("eyoc_spconv_upc_min_rows", prev_min)
        _lib.knob("eyoc_maps_lazy_tables", 1)


def test_fused_level_construction_equals_the_per_level_kernels():
    """Round 6: a Z-ordered build makes the three coarser levels - coordinates, parent and child links - in two launches over the sorted
    level-0 rows (coordmap.hip k_levels_count / k_levels_fill) instead of flag / scan / scan / compact per level.  Everything derived
    from them must be what the per-level kernels give, bit for bit: level coordinates, every table of every level (the Z-ordered build
    derives them top-down from the links), the pair counts, the forward.  Sizes: a batch of two clouds, one small cloud forced into
    Z-order, row counts around the 2048-row tile and the 64-row wave step, a 3-row cloud, one voxel."""
    import eyoc_amd
    from eyoc_amd import _lib, synthetic as syn
    from oracle import coords as oc
    from test_gpu_round2 import _model
    rng = np.random.default_rng(17)
    p = syn.make_pair(5)
    cases = [syn.batch_coords([p["coords0"], p["coords1"]])]
    for n in (2048, 2049, 4095, 63, 64, 65, 3, 1):
        c = np.unique(rng.integers(-40, 40, size=(4 * n + 8, 3)), axis=0)[:n].astype(np.int32)
        cases.append(syn.batch_coords([c]))
    cases.append(syn.batch_coords([np.unique(rng.integers(-9, 9, size=(900, 3)), axis=0).astype(np.int32) for _ in range(5)]))   # five dense clouds
    far = np.unique(rng.integers(-65000, 65000, size=(3000, 3)), axis=0).astype(np.int32)            # every row its own voxel at every level, both signs, 17-bit coordinates
    near = (np.unique(rng.integers(0, 12, size=(400, 3)), axis=0) + np.array([-64990, 64900, -7])).astype(np.int32)   # a dense patch across sign and power-of-two boundaries
    cases.append(syn.batch_coords([np.concatenate([far, near]), near - np.array([0, 129000, 0], np.int32)]))
    model, _sd = _model()
    prev_order = _lib.knob("eyoc_maps_internal_order", 1) - 2
    assert _lib.knob("eyoc_maps_fused_levels", -7) == 1
    try:
        for ci, coords in enumerate(cases):
            got = []
            for fused in (1, 0):
                _lib.knob("eyoc_maps_fused_levels", fused)
                feats = np.ones((len(coords), 1), np.float32)
                x = eyoc_amd.SparseTensor(torch.from_numpy(feats).cuda(), coordinates=torch.from_numpy(coords).cuda())
                cm = x.coordinate_manager
                cm.maps(-1)
                rec = {"order": cm.row_order().cpu().numpy(), "info": cm.info(conv1_kernel_size=5)}
                for l in range(4):
                    rec[f"c{l}"] = cm.level_coordinates(l, internal=True).cpu().numpy()
                    rec[f"s1_{l}"] = cm.table(_lib.MAP_S1, l, internal=True).cpu().numpy()
                    if l < 3:
                        rec[f"up_{l}"] = cm.table(_lib.MAP_UP, l, internal=True).cpu().numpy()
                        rec[f"down_{l}"] = cm.table(_lib.MAP_DOWN, l, internal=True).cpu().numpy()
                if len(coords) >= 64:
                    rec["F"] = model(x).F.cpu().numpy()
                got.append(rec)
            a, b = got
            assert a["info"] == b["info"], ci
            for k in a:
                if k != "info":
                    np.testing.assert_array_equal(a[k], b[k], err_msg=f"case {ci} ({len(coords)} rows), {k}")
            want = oc.build_maps(coords[a["order"]])
            for l in range(4):
                np.testing.assert_array_equal(a[f"s1_{l}"], want["s1"][l], err_msg=f"case {ci}, s1 level {l} against the oracle")
    finally:
        _lib.knob("eyoc_maps_fused_levels", 1)
        _lib.knob("eyoc_maps_internal_order", prev_order)
