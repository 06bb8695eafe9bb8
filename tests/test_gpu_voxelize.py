"""GPU parity: hash-grid voxeliser vs the oracle - integer work, bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def check(xyz, voxel, batch=0):
    import eyoc_amd
    from oracle import voxelize as ov
    coords, sel = eyoc_amd.sparse_quantize(xyz, voxel, batch)
    rc, rs = ov.sparse_quantize(xyz, voxel, batch)
    np.testing.assert_array_equal(sel.cpu().numpy(), rs)
    np.testing.assert_array_equal(coords.cpu().numpy(), rc)
    return coords, sel


def test_voxelize_raw_lidar_sweep_matches_generator():
    from eyoc_amd import synthetic as syn
    scene = syn.make_scene(np.random.default_rng(5))
    pts = syn.raycast(scene, syn._pose(0, 0, 0), np.random.default_rng(1), brush_level=1.0)
    assert len(pts) > 100000
    coords, sel = check(pts, 0.3)
    s2, c2 = syn.voxelize(pts, 0.3)                       # the generator's own voxeliser
    np.testing.assert_array_equal(sel.cpu().numpy(), s2)
    np.testing.assert_array_equal(coords.cpu().numpy()[:, 1:], c2)
    xyzr = np.concatenate([pts, np.zeros((len(pts), 1), np.float32)], 1)     # KITTI .bin layout
    c4, s4 = check(xyzr, 0.3, batch=3)
    assert (c4[:, 0] == 3).all() and torch.equal(s4, sel)


def test_voxelize_edge_cases():
    import eyoc_amd
    rng = np.random.default_rng(0)
    check(rng.uniform(-3, 3, (5000, 3)).astype(np.float32), 0.5)          # heavy duplication, negative cells
    check(np.array([[0.0, 0.0, 0.0]], np.float32), 0.3)
    check(np.array([[-1e-7, 0.29999, 0.3], [-0.3, 0.3, 0.6], [-0.30001, 0.0, 0.0]], np.float32), 0.3)   # cell borders
    c, s = eyoc_amd.sparse_quantize(np.zeros((0, 3), np.float32), 0.3)
    assert c.shape == (0, 4) and s.shape == (0,)
    with pytest.raises(eyoc_amd.EyocError, match="range"):
        eyoc_amd.sparse_quantize(np.array([[1e6, 0, 0]], np.float32), 0.3)


def test_extract_features_one_call_api():
    import eyoc_amd
    from eyoc_amd import synthetic as syn
    from oracle import resunet as orr, voxelize as ov
    rng = np.random.default_rng(2)
    xyz = rng.uniform(-4, 4, (6000, 3)).astype(np.float32)
    sd = syn.make_weights()
    model = eyoc_amd.load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, conv1_kernel_size=5, normalize_feature=True)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model = model.cuda()
    pts, F = eyoc_amd.extract_features(model, xyz, voxel_size=0.3, device=torch.device("cuda:0"))
    coords, sel = ov.sparse_quantize(xyz, 0.3)
    np.testing.assert_array_equal(pts, xyz[sel])
    ref = orr.resunet_forward(sd, coords, np.ones((len(sel), 1), np.float32)).numpy()
    assert np.abs(F.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max()
