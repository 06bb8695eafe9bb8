"""GPU parity: pose solvers (Kabsch, IRLS, RANSAC) through the C ABI vs the oracle and vs the
reference's golden vectors.  Floating point tolerance: 1e-4 on pose entries (north_star)."""
import json
import os

import numpy as np
import pytest
import torch

import _inputs as gi

pytestmark = pytest.mark.gpu


def _golden(name):
    return np.load(os.path.join(os.path.dirname(__file__), "golden", name))


def test_rigid_transform_3d_vs_golden():
    import eyoc_amd
    from make_golden_inputs import kabsch_inputs
    from test_oracle_golden import well_conditioned
    g = _golden("g3_kabsch.npz")
    for i, case in enumerate(json.loads(str(g["cases"]))):
        A, B, w = kabsch_inputs(*case)
        T = eyoc_amd.rigid_transform_3d(torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda(),
                                        None if w is None else torch.from_numpy(w).cuda())
        assert T.is_cuda and T.shape == (A.shape[0], 4, 4)
        T = T.cpu().numpy()
        ok = well_conditioned(A, B, w)
        print(f"kabsch golden case {i} {case}: max |T - T_ref| = {np.abs(T[ok] - g[f'T{i}'][ok]).max():.2e} over {int(ok.sum())} problems")
        np.testing.assert_allclose(T[ok], g[f"T{i}"][ok], rtol=0, atol=1e-4, err_msg=f"case {i} {case}")   # realised: <= 1.6e-5
        np.testing.assert_allclose(np.linalg.det(T[:, :3, :3]), 1.0, atol=1e-5)
        np.testing.assert_array_equal(T[:, 3], np.tile([0, 0, 0, 1], (len(T), 1)))
    # CPU inputs are uploaded and the result comes back on the CPU
    Tc = eyoc_amd.rigid_transform_3d(torch.from_numpy(A), torch.from_numpy(B), None)
    assert not Tc.is_cuda


def test_rigid_transform_degenerate_inputs():
    import eyoc_amd
    A = torch.rand(3, 10, 3).cuda()
    w0 = torch.zeros(3, 10).cuda()
    T = eyoc_amd.rigid_transform_3d(A, A + 1.0, w0).cpu().numpy()       # no weight at all -> identity rotation
    np.testing.assert_allclose(T[:, :3, :3], np.tile(np.eye(3), (3, 1, 1)), atol=1e-6)
    line = torch.stack([torch.linspace(0, 1, 10)] * 3, 1)[None].cuda()   # collinear points: still a rotation
    T = eyoc_amd.rigid_transform_3d(line, line, None).cpu().numpy()
    assert np.isfinite(T).all() and abs(np.linalg.det(T[0, :3, :3]) - 1) < 1e-5


def test_est_quad_linear_robust_vs_golden():
    import eyoc_amd
    g = _golden("g2_irls.npz")
    for i, (seed, n, frac, use_w, tp) in enumerate(json.loads(str(g["cases"]))):
        p0, p1, _ = gi.corr_case(seed, n, gi.rigid(*tp), frac)
        w = None
        if use_w:
            w = torch.from_numpy((0.05 + 0.95 * gi._u(seed + 9, n, 1)).astype(np.float32))
        T = eyoc_amd.est_quad_linear_robust(torch.from_numpy(p0), torch.from_numpy(p1), w)
        assert T.shape == (4, 4) and not T.is_cuda          # CPU in -> CPU out, like the reference
        np.testing.assert_allclose(T.numpy(), g[f"T{i}"], rtol=0, atol=1e-4, err_msg=f"case {i}")
    assert eyoc_amd.estimate_transform is eyoc_amd.est_quad_linear_robust


def test_ransac_matches_oracle_hypothesis_for_hypothesis():
    import eyoc_amd
    from oracle import ransac as orn
    T = gi.rigid(0.01, -0.02, 0.15, 9.0, 0.5, 0.1)
    for seed, n, frac, H in ((61, 2000, 0.3, 200000), (62, 5000, 0.15, 400000)):
        p0, p1, inl = gi.corr_case(seed, n, T, frac, noise=0.03)
        corr = torch.arange(n, dtype=torch.int64)
        res = eyoc_amd.ransac_from_correspondences(torch.from_numpy(p0), torch.from_numpy(p1), corr, 0.3, H, seed=3)
        ref = orn.ransac(p0, p1, np.arange(n), 0.3, H, seed=3)
        assert res.survivors == ref["survivors"]
        assert res.best_hypothesis == ref["best_h"]
        assert res.inliers == ref["inliers"]
        assert res.fitness == pytest.approx(ref["fitness"])
        assert res.inlier_rmse == pytest.approx(ref["rmse"], rel=1e-5)
        np.testing.assert_allclose(res.transformation, ref["T"], atol=1e-5)
        np.testing.assert_allclose(res.transformation[:3, :3], T[:3, :3], atol=0.03)


def test_ransac_no_survivor_and_permuted_correspondences():
    import eyoc_amd
    from oracle import ransac as orn
    rng = np.random.default_rng(0)
    p0 = rng.uniform(-30, 30, (500, 3)).astype(np.float32)
    p1 = rng.uniform(-30, 30, (700, 3)).astype(np.float32)          # pure outliers
    corr = torch.from_numpy(rng.integers(0, 700, 500))
    res = eyoc_amd.ransac_from_correspondences(torch.from_numpy(p0), torch.from_numpy(p1), corr, 0.05, 20000, seed=1)
    ref = orn.ransac(p0, p1, corr.numpy(), 0.05, 20000, seed=1)
    assert res.survivors == ref["survivors"]
    if ref["survivors"] == 0:
        assert res.best_hypothesis == -1 and res.inliers == 0
        np.testing.assert_array_equal(res.transformation, np.eye(4))


def test_ransac_batched_equals_per_pair_and_oracle():
    """Ragged batch of 11 pairs (more than one launch chunk), one pair too large for the LDS-staged records:
    every result is bit-identical to the single-pair call with seed + b, and the small ones to the oracle."""
    import eyoc_amd
    from eyoc_amd import registration as reg
    from oracle import ransac as orn
    T = gi.rigid(0.02, 0.01, -0.12, 5.0, -0.3, 0.2)
    sizes = [400, 1500, 7000, 5, 900, 2048, 33, 640, 1200, 777, 3100]
    H = 60000
    src, tgt, corr, seg_s, seg_t, cases = [], [], [], [0], [0], []
    for b, n in enumerate(sizes):
        p0, p1, _ = gi.corr_case(200 + b, n, T, 0.25 if n > 100 else 1.0, noise=0.03)
        perm = np.random.default_rng(b).permutation(n)
        extra = np.random.default_rng(100 + b).uniform(-20, 20, (b * 3, 3)).astype(np.float32)   # unmatched target rows
        t_rows = np.concatenate([p1[perm], extra])
        c = np.argsort(perm)                                # source i matches target row c[i]
        src.append(p0); tgt.append(t_rows); corr.append(c)
        seg_s.append(seg_s[-1] + n); seg_t.append(seg_t[-1] + len(t_rows))
        cases.append((p0, t_rows, c))
    res = reg.ransac_batched_from_correspondences(torch.from_numpy(np.concatenate(src)), torch.from_numpy(np.concatenate(tgt)),
                                                  torch.from_numpy(np.concatenate(corr)), seg_s, seg_t, 0.3, H, seed=40).cpu()
    for b, (p0, t_rows, c) in enumerate(cases):
        got = reg.decode_ransac_result(res[b], len(p0))
        one = eyoc_amd.ransac_from_correspondences(torch.from_numpy(p0), torch.from_numpy(t_rows), torch.from_numpy(c), 0.3, H,
                                                   seed=40 + b)
        assert (got.survivors, got.best_hypothesis, got.inliers) == (one.survivors, one.best_hypothesis, one.inliers), b
        np.testing.assert_array_equal(got.transformation, one.transformation)
        if len(p0) <= 1500:
            ref = orn.ransac(p0, t_rows, c, 0.3, H, seed=40 + b)
            assert (got.survivors, got.best_hypothesis, got.inliers) == (ref["survivors"], ref["best_h"], ref["inliers"]), b
            np.testing.assert_allclose(got.transformation, ref["T"], atol=1e-5)


def test_feature_matching_ransac_end_to_end():
    import eyoc_amd
    T = gi.rigid(0.0, 0.01, -0.1, 6.0, -0.4, 0.05)
    p0, p1, inl = gi.corr_case(71, 3000, T, 1.0, noise=0.02)
    F0 = gi.unit_feats(72, 3000)
    perm = np.random.default_rng(1).permutation(3000)
    F1 = F0[perm].copy()
    F1[::3] = gi.unit_feats(73, 3000)[::3]                           # a third of the descriptors are junk
    res = eyoc_amd.registration_ransac_based_on_feature_matching(p0, p1[perm], torch.from_numpy(F0),
                                                                 torch.from_numpy(F1), False, 0.3,
                                                                 criteria=(100000, 10000), seed=5)
    rte, rre, ok = eyoc_amd.registration_errors(res.transformation, T)
    assert ok and rte < 0.1 and rre < np.deg2rad(0.5)
    assert res.fitness > 0.5


def test_open3d_shaped_call_site_runs_unchanged():
    """The reference's RANSAC call site in its own shape (scripts/test_kitti.py:159-177 with util/pointcloud.py:9-21 building
    the arguments): ``PointCloud`` + ``Vector3dVector``, ``Feature.resize`` + ``[C, n]`` float64 ``data``, the estimation /
    checker / criteria objects, ``result.transformation``.  Same answer as the array-level entry point."""
    import eyoc_amd
    import eyoc_amd.o3d as o3d
    T = gi.rigid(0.0, 0.01, -0.1, 6.0, -0.4, 0.05)
    n = 3000
    xyz0np, xyz1np, _ = gi.corr_case(71, n, T, 1.0, noise=0.02)
    F0 = torch.from_numpy(gi.unit_feats(72, n)).cuda()
    perm = np.random.default_rng(1).permutation(n)
    F1 = F0[torch.from_numpy(perm).cuda()].clone()
    F1[::3] = torch.from_numpy(gi.unit_feats(73, n)[::3]).cuda()
    xyz1np = xyz1np[perm]

    def make_open3d_point_cloud(xyz):
        pcd = o3d.geometry.PointCloud()
        pcd.points = o3d.utility.Vector3dVector(xyz)
        return pcd

    def make_open3d_feature(data, dim, npts):
        feature = o3d.pipelines.registration.Feature()
        feature.resize(dim, npts)
        feature.data = data.cpu().numpy().astype('d').transpose()
        return feature
    pcd0, pcd1 = make_open3d_point_cloud(xyz0np), make_open3d_point_cloud(xyz1np)
    feat0, feat1 = make_open3d_feature(F0, 32, F0.shape[0]), make_open3d_feature(F1, 32, F1.shape[0])
    distance_threshold = 0.3
    ransac_result = o3d.pipelines.registration.registration_ransac_based_on_feature_matching(
        pcd0, pcd1, feat0, feat1, False, distance_threshold,
        o3d.pipelines.registration.TransformationEstimationPointToPoint(False), 4, [
            o3d.pipelines.registration.CorrespondenceCheckerBasedOnEdgeLength(0.9),
            o3d.pipelines.registration.CorrespondenceCheckerBasedOnDistance(distance_threshold)
        ], o3d.pipelines.registration.RANSACConvergenceCriteria(100000, 10000))
    T_ransac = torch.from_numpy(ransac_result.transformation.astype(np.float32))
    direct = eyoc_amd.registration_ransac_based_on_feature_matching(xyz0np, xyz1np, F0, F1, False, 0.3, criteria=(100000, 10000))
    np.testing.assert_array_equal(ransac_result.transformation, direct.transformation)
    rte, rre, ok = eyoc_amd.registration_errors(T_ransac.numpy(), T)
    assert ok and rte < 0.1 and ransac_result.fitness > 0.5
    with pytest.raises(NotImplementedError):
        o3d.pipelines.registration.registration_ransac_based_on_feature_matching(
            pcd0, pcd1, feat0, feat1, False, distance_threshold,
            o3d.pipelines.registration.TransformationEstimationPointToPoint(True), 4, [], o3d.pipelines.registration.RANSACConvergenceCriteria(1000, 1.0))
