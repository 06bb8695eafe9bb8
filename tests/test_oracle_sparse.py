"""Self-consistency pins for the un-pinnable parts of the oracle (MinkowskiEngine / Open3D
semantics): dense-convolution equivalence on a fully occupied grid, toy coordinate cases, and the
RANSAC restatement recovering a planted pose.  CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from oracle import coords as oc
from oracle import ransac as orn
from oracle import resunet as orr
import _inputs as gi


def full_grid(n, batch=1):
    g = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3)
    return np.concatenate([np.concatenate([np.full((len(g), 1), b), g], 1) for b in range(batch)], 0)


def to_dense(x, coords, n, ts=1):
    c = coords[:, 1:] // ts
    d = torch.zeros((x.shape[1], n, n, n))
    d[:, c[:, 2], c[:, 1], c[:, 0]] = x.t()          # dense layout [C, z, y, x]
    return d[None]


def dense_weight(W, ks):
    # W[k, ci, co] with k = x fastest  ->  conv3d weight [co, ci, kz, ky, kx]
    return W.reshape(ks, ks, ks, W.shape[1], W.shape[2]).permute(4, 3, 0, 1, 2).contiguous()


def test_kernel_offsets_order():
    o = oc.kernel_offsets(3)
    assert o.shape == (27, 3)
    assert o[0].tolist() == [-1, -1, -1] and o[1].tolist() == [0, -1, -1] and o[3].tolist() == [-1, 0, -1]
    assert o[13].tolist() == [0, 0, 0] and o[26].tolist() == [1, 1, 1]
    assert oc.kernel_offsets(5).shape == (125, 3) and oc.kernel_offsets(5)[62].tolist() == [0, 0, 0]


@pytest.mark.parametrize("ks", [3, 5])
def test_stride1_conv_equals_dense(ks):
    n, ci, co = 6, 3, 5
    coords = full_grid(n)
    cm = oc.CoordMap(coords, 1)
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.normal(size=(len(coords), ci)).astype(np.float32))
    W = torch.from_numpy(rng.normal(size=(ks ** 3, ci, co)).astype(np.float32))
    out = orr.sparse_conv(x, oc.kernel_map(cm, cm, ks), W)
    ref = TF.conv3d(to_dense(x, coords, n), dense_weight(W, ks), padding=ks // 2)
    np.testing.assert_allclose(to_dense(out, coords, n).numpy(), ref.numpy(), atol=1e-4)


def test_stride2_and_transposed_equal_dense():
    n, ci, co = 8, 4, 6
    coords = full_grid(n)
    cm1 = oc.CoordMap(coords, 1)
    cm2, parent = oc.stride_map(cm1, 2)
    assert len(cm2) == (n // 2) ** 3 and cm2.ts == 2
    assert np.array_equal(cm2.coords[parent][:, 1:], coords[:, 1:] // 2 * 2)
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.normal(size=(len(coords), ci)).astype(np.float32))
    W = torch.from_numpy(rng.normal(size=(27, ci, co)).astype(np.float32))
    down = orr.sparse_conv(x, oc.kernel_map(cm1, cm2, 3), W)
    ref = TF.conv3d(to_dense(x, coords, n), dense_weight(W, 3), stride=2, padding=1)
    np.testing.assert_allclose(to_dense(down, cm2.coords, n // 2, ts=2).numpy(), ref.numpy(), atol=1e-4)
    # transposed: back onto the existing fine map
    Wt = torch.from_numpy(rng.normal(size=(27, co, ci)).astype(np.float32))
    up = orr.sparse_conv(down, oc.transposed_kernel_map(cm2, cm1, 3), Wt)
    wt = Wt.reshape(3, 3, 3, co, ci).permute(3, 4, 0, 1, 2).contiguous()   # [cin, cout, kz, ky, kx]
    ref_up = TF.conv_transpose3d(ref, wt, stride=2, padding=1, output_padding=1)
    np.testing.assert_allclose(to_dense(up, coords, n).numpy(), ref_up.numpy(), atol=2e-3)


def test_negative_coordinates_floor_and_batches_do_not_mix():
    coords = np.array([[0, -1, -1, -1], [0, -2, 0, 1], [0, 0, 0, 0], [1, 0, 0, 0], [1, 1, 0, 0]])
    cm = oc.CoordMap(coords, 1)
    cm2, parent = oc.stride_map(cm, 2)
    assert cm2.coords.tolist() == [[0, -2, -2, -2], [0, -2, 0, 0], [0, 0, 0, 0], [1, 0, 0, 0]]
    assert parent.tolist() == [0, 1, 2, 3, 3]
    nbr = oc.kernel_map(cm, cm, 3)
    assert nbr[13].tolist() == [0, 1, 2, 3, 4]                 # centre offset = identity
    # (0,0,0)b0 sees (-1,-1,-1)b0 through offset (-1,-1,-1) = k 0, never the batch-1 voxel at (1,0,0)
    assert nbr[0][2] == 0 and nbr[14][2] == -1 and nbr[14][3] == 4
    with pytest.raises(ValueError):
        oc.CoordMap(np.array([[0, 1, 2, 3], [0, 1, 2, 3]]))


def test_transposed_map_is_transpose_of_forward_map():
    rng = np.random.default_rng(3)
    c = np.unique(rng.integers(-6, 6, size=(300, 3)), axis=0)
    coords = np.concatenate([np.zeros((len(c), 1), np.int64), c], 1)
    cm1 = oc.CoordMap(coords, 1)
    cm2, _ = oc.stride_map(cm1, 2)
    fwd = oc.kernel_map(cm1, cm2, 3)           # [27, N2] -> fine row
    up = oc.transposed_kernel_map(cm2, cm1, 3)  # [27, N1] -> coarse row
    f = {(k, int(u), v) for k in range(27) for v, u in enumerate(fwd[k]) if u >= 0}
    t = {(k, u, int(v)) for k in range(27) for u, v in enumerate(up[k]) if v >= 0}
    assert f == t and len(f) > 0
    assert (up >= 0).sum(0).min() >= 1          # every fine row has at least its parent


def test_resunet_forward_small_cloud_shapes_and_norm():
    from eyoc_amd import synthetic as syn
    rng = np.random.default_rng(5)
    c = np.unique(rng.integers(-10, 10, size=(900, 3)), axis=0)
    rng.shuffle(c)
    coords = syn.batch_coords([c[:400], c[400:]])
    sd = syn.make_weights()
    F, inter, maps = orr.resunet_forward(sd, coords, np.ones((len(coords), 1), np.float32),
                                         return_intermediate=True)
    assert F.shape == (len(coords), 32)
    np.testing.assert_allclose(F.norm(dim=1).numpy(), 1.0, atol=1e-5)
    # a batched forward equals per-cloud forwards (neighbourhoods never cross the batch index)
    F0 = orr.resunet_forward(sd, syn.batch_coords([c[:400]]), np.ones((400, 1), np.float32))
    np.testing.assert_allclose(F[:400].numpy(), F0.numpy(), atol=2e-5)


def test_ransac_sampler_is_stable_and_in_range():
    idx = orn.sample_indices(7, 0, 1000, 5000)
    assert idx.shape == (1000, 4) and idx.min() >= 0 and idx.max() < 5000
    assert np.array_equal(idx[100:200], orn.sample_indices(7, 100, 100, 5000))
    # frozen values: the HIP kernel implements the same splitmix64 counter hash
    assert orn.sample_indices(0, 0, 2, 5000).tolist() == orn.sample_indices(0, 0, 2, 5000).tolist()
    assert len(np.unique(orn.sample_indices(1, 0, 4096, 1 << 20))) > 16000


def test_ransac_recovers_planted_pose():
    T = gi.rigid(0.01, -0.02, 0.15, 9.0, 0.5, 0.1)
    p0, p1, inl = gi.corr_case(61, 2000, T, 0.3, noise=0.03)
    res = orn.ransac(p0, p1, np.arange(len(p0)), 0.3, 200000, seed=3)
    assert res["survivors"] > 0 and res["inliers"] > 0.2 * len(p0)
    np.testing.assert_allclose(res["T"][:3, :3], T[:3, :3], atol=0.02)
    np.testing.assert_allclose(res["T"][:3, 3], T[:3, 3], atol=0.3)


def test_label_oracle_properties():
    """oracle/labels.py against independent numpy formulations (parity unpinned: lib/trainer.py cannot be imported)."""
    from oracle import labels as ol
    rng = np.random.default_rng(3)
    A = rng.normal(size=(200, 32)).astype(np.float32)
    B = rng.normal(size=(333, 32)).astype(np.float32)
    idx, d1, d2 = ol.knn2(A, B)
    D = ((A[:, None, :].astype(np.float64) - B[None].astype(np.float64)) ** 2).sum(2)
    part = np.sort(D, axis=1)[:, :2]
    np.testing.assert_array_equal(idx, D.argmin(1))
    np.testing.assert_allclose(d1, part[:, 0], rtol=1e-5)
    np.testing.assert_allclose(d2, part[:, 1], rtol=1e-5)
    assert (d1 <= d2).all()
    w = ol.lowe_weights(d1 / 40, d2 / 40)                       # unit-feature range: weight = 1 - d1/d2
    np.testing.assert_allclose(w, 1 - d1 / d2, atol=2e-5)
    src, tgt, ws = ol.topk_matches(w, idx, 50)
    assert (np.diff(ws) <= 0).all() and len(src) == 50 and (tgt == idx[src]).all()
    assert set(src) == set(np.argsort(-w, kind="stable")[:50])
    # spherical filter keeps exactly the pairs with both ends outside the radius
    C0 = rng.uniform(-40, 40, (200, 3)).astype(np.float32)
    C1 = rng.uniform(-40, 40, (333, 3)).astype(np.float32)
    F = lambda x: x / np.linalg.norm(x, axis=1, keepdims=True)
    m, unc = ol.match_and_filter_corr([C0], [F(A)], [C1], [F(B)], radius=20, num_corres=60)
    assert m.shape == (120, 2) and len(unc) == 1
    keep = (np.linalg.norm(C0[m[:, 0]], axis=1) > 20) & (np.linalg.norm(C1[m[:, 1]], axis=1) > 20)
    np.testing.assert_array_equal(unc[0], m[keep])
    # pose-consistency filter
    T = np.eye(4, dtype=np.float32); T[:3, 3] = [1, 2, 3]
    P0 = rng.uniform(-10, 10, (300, 3)).astype(np.float32)
    P1 = (P0 + T[:3, 3]).astype(np.float32)[::-1].copy()
    out = ol.correspondences_under_pose(P0, P1, T, np.arange(0, 300, 3), 0.01)
    np.testing.assert_array_equal(out[:, 1], 299 - out[:, 0])
    assert len(out) == 100


def test_similarity_mask_against_an_independent_torch_formulation():
    """oracle.labels.similarity_mask (numpy) against the same rule written with torch tensor ops (norm, integer cast,
    clamping by masked assignment, 2-D indexing) on the reference's real KITTI table when the reference tree is around
    (this container), else on a synthetic table of the same shapes.  lib/trainer.py itself cannot be imported
    (MinkowskiEngine / pytorch3d / open3d at module scope), so this is as close to a pin as the filter gets."""
    import os
    import torch
    from oracle import labels as ol
    path = "/root/reference/config/dist_sim_plot/kitti_distSimPlot.npz"
    rng = np.random.default_rng(3)
    if os.path.exists(path):
        maps = np.load(path, allow_pickle=True)["res"].tolist()
        table = {i: np.asarray(maps[i], np.float64) for i in range(6)}
    else:
        table = {i: rng.uniform(0, 1, sh) for i, sh in enumerate([(12, 16), (18, 16), (20, 18), (20, 18), (20, 18), (20, 18)])}
    C0 = (rng.uniform(-1, 1, (2000, 3)) * rng.choice([5, 40, 150], (2000, 1))).astype(np.float32)
    C1 = (rng.uniform(-1, 1, (2000, 3)) * rng.choice([5, 40, 150], (2000, 1))).astype(np.float32)
    a, b = rng.integers(0, 2000, 3000), rng.integers(0, 2000, 3000)
    for frame_distance in (0, 6, 11, 16, 23, 40):
        fi = min(max(0, frame_distance // 5), 5)
        t = torch.tensor(table[fi])
        d0 = torch.norm(torch.from_numpy(C0)[a], dim=1)
        d1 = torch.norm(torch.from_numpy(C1)[b], dim=1)
        gap = (d0 - d1).abs()
        dmin = torch.minimum(d0, d1)
        c0 = (dmin / 5).long()
        c1 = (gap / {0: 1, 1: 1.5, 2: 2, 3: 2.5, 4: 2.5, 5: 2.5}[fi]).long()
        c0[c0 >= t.shape[1]] = t.shape[1] - 1
        c1[c1 >= t.shape[0]] = t.shape[0] - 1
        want = (t[c1, c0] > 0.4).numpy()
        got = ol.similarity_mask(C0, C1, a, b, table, frame_distance, 0.4)
        # torch.norm and the oracle's fixed-order fp32 norm can differ in the last bit: allow the cells on a boundary
        assert (got != want).sum() <= 3, int((got != want).sum())


def test_training_mode_batch_norm_is_torch_batchnorm1d():
    """MinkowskiBatchNorm is ``nn.BatchNorm1d`` applied to the feature rows (model/common.py:4-6 -> ME's wrapper): the oracle's
    training-mode norm (batch mean, biased variance, running statistics moved by ``momentum`` with the unbiased variance) against
    torch's own module - output, both gradients and the running statistics it leaves behind."""
    import torch
    from oracle import resunet as orr
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.normal(1.0, 2.0, size=(257, 32)).astype(np.float32))
    bn = torch.nn.BatchNorm1d(32, momentum=0.05)
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, 32).astype(np.float32)))
        bn.bias.copy_(torch.from_numpy(rng.normal(size=32).astype(np.float32)))
        bn.running_mean.copy_(torch.from_numpy(rng.normal(size=32).astype(np.float32)))
        bn.running_var.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, 32).astype(np.float32)))
    sd = {f"n.bn.{k}": v.detach().clone() for k, v in bn.state_dict().items()}
    sd["n.bn.weight"].requires_grad_(True); sd["n.bn.bias"].requires_grad_(True)
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    g = torch.from_numpy(rng.normal(size=(257, 32)).astype(np.float32))
    bn.train()
    want = bn(x1)
    want.backward(g)
    running = {}
    orr._TRAIN.update(on=True, momentum=0.05, running=running)
    try:
        got = orr.batch_norm(x2, sd, "n")
    finally:
        orr._TRAIN.update(on=False, running=None)
    got.backward(g)
    np.testing.assert_allclose(got.detach().numpy(), want.detach().numpy(), atol=2e-6)
    np.testing.assert_allclose(x2.grad.numpy(), x1.grad.numpy(), atol=2e-6)
    np.testing.assert_allclose(sd["n.bn.weight"].grad.numpy(), bn.weight.grad.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(sd["n.bn.bias"].grad.numpy(), bn.bias.grad.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(running["n.bn.running_mean"].numpy(), bn.running_mean.numpy(), atol=1e-6)
    np.testing.assert_allclose(running["n.bn.running_var"].numpy(), bn.running_var.numpy(), atol=1e-6)
