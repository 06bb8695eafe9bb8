"""GPU parity: SC2-PCR (Matcher) through the C ABI vs the reference's golden poses and vs the oracle."""
import json
import os

import numpy as np
import pytest
import torch

import _inputs as gi

pytestmark = pytest.mark.gpu


def _golden(name):
    return np.load(os.path.join(os.path.dirname(__file__), "golden", name))


def test_sc2pcr_matches_reference_golden_poses():
    import eyoc_amd
    from oracle import sc2pcr as osc
    g = _golden("g4_sc2pcr.npz")
    cfg = json.loads(str(g["cfg"]))
    m = eyoc_amd.Matcher(**cfg)
    mo = osc.Matcher(**cfg)
    for i, (seed, n, frac, tp) in enumerate(json.loads(str(g["cases"]))):
        p0, p1, _ = gi.corr_case(seed, n, gi.rigid(*tp), frac, noise=0.03)
        T, fit = m.SC2_PCR(torch.from_numpy(p0)[None].cuda(), torch.from_numpy(p1)[None].cuda())
        assert T.shape == (1, 4, 4) and fit.shape == (1, int(n * cfg["ratio"]))
        print(f"sc2pcr golden case {i}: max |T - T_ref| = {np.abs(T[0].cpu().numpy() - g[f'T{i}']).max():.2e}")
        np.testing.assert_allclose(T[0].cpu().numpy(), g[f"T{i}"], rtol=0, atol=1e-4, err_msg=f"case {i}")   # realised: <= 3e-5
        assert float(fit.max()) == pytest.approx(float(g[f"fitmax{i}"]), abs=2)
        # seed-wise fitness: same multiset of hypotheses as the oracle up to tie-breaking noise
        To, fo = mo.SC2_PCR(torch.from_numpy(p0)[None], torch.from_numpy(p1)[None])
        assert abs(float(fit.sum()) - float(fo.sum())) <= 0.02 * float(fo.sum()) + 10


def test_sc2pcr_small_inputs_and_estimator():
    import eyoc_amd
    from oracle import sc2pcr as osc
    T = gi.rigid(0.01, 0.0, 0.08, 3.0, 0.2, 0.0)
    p0, p1, _ = gi.corr_case(81, 20, T, 1.0, noise=0.01)          # k1 > n -> k1 = k2 = 4 branch
    m = eyoc_amd.Matcher(**osc.KITTI_CFG)
    Tg, fit = m.SC2_PCR(torch.from_numpy(p0)[None].cuda(), torch.from_numpy(p1)[None].cuda())
    To, _ = osc.Matcher(**osc.KITTI_CFG).SC2_PCR(torch.from_numpy(p0)[None], torch.from_numpy(p1)[None])
    print(f"sc2pcr n=20 vs oracle: max |T - T_oracle| = {np.abs(Tg[0].cpu().numpy() - To[0].numpy()).max():.2e}")
    np.testing.assert_allclose(Tg[0].cpu().numpy(), To[0].numpy(), atol=1e-4)   # realised: 1.4e-6
    # estimator: descriptors that identify the correspondence exactly
    n = 600
    q0, q1, _ = gi.corr_case(82, n, T, 1.0, noise=0.01)
    F = gi.unit_feats(83, n)
    perm = np.random.default_rng(0).permutation(n)
    mm = eyoc_amd.Matcher(**{**osc.KITTI_CFG, "num_node": "all"})
    out = mm.estimator(torch.from_numpy(q0)[None].cuda(), torch.from_numpy(q1[perm])[None].cuda(),
                       torch.from_numpy(F)[None].cuda(), torch.from_numpy(F[perm])[None].cuda())
    Te, labels, sc, tc, fitness = out
    np.testing.assert_allclose(Te[0].cpu().numpy(), T, atol=0.02)
    assert labels.shape == (1, n) and labels.mean() > 0.95
    np.testing.assert_allclose(tc[0].cpu().numpy(), q1, atol=1e-6)   # matching undid the permutation
    with pytest.raises(NotImplementedError):
        mm.SC2_PCR(torch.zeros(2, 10, 3).cuda(), torch.zeros(2, 10, 3).cuda())


def test_sc2pcr_kitti_sized_problem_recovers_pose():
    """N = 8000 correspondences as produced by match_pair's resampling (37 % duplicates), 25 % inliers."""
    import eyoc_amd
    from oracle import sc2pcr as osc
    T = gi.rigid(0.0, 0.01, 0.1, 11.0, -0.3, 0.05)
    p0, p1, inl = gi.corr_case(91, 5000, T, 0.25, noise=0.03)
    sel = np.random.default_rng(1).choice(5000, 8000)
    m = eyoc_amd.Matcher(**osc.KITTI_CFG)
    Tg, fit = m.SC2_PCR(torch.from_numpy(p0[sel])[None].cuda(), torch.from_numpy(p1[sel])[None].cuda())
    rte, rre, ok = eyoc_amd.registration_errors(Tg[0].cpu().numpy(), T)
    assert ok and rte < 0.05 and rre < np.deg2rad(0.2)
    assert fit.shape == (1, 1600)


def test_sc2pcr_batched_equals_per_pair():
    """19 ragged pairs (more than one 16-pair launch chunk): every pose and seedwise fitness is bit-identical to the
    single-pair call, and the caller's stream sees the results without an explicit sync."""
    import time
    import eyoc_amd
    m = eyoc_amd.Matcher(inlier_threshold=0.6, d_thre=0.1, ratio=0.2, nms_radius=0.6, max_points=8000, k1=30, k2=20,
                         num_iterations=20)
    T_gt = gi.rigid(0.02, -0.01, 0.1, 4.0, 0.3, -0.2)
    sizes = [1200, 3000, 64, 800, 2500, 5000, 333, 1500, 2048, 900, 4100, 700, 1999, 256, 3100, 1000, 40, 2222, 650]
    src, tgt = [], []
    for b, n in enumerate(sizes):
        p0, p1, _ = gi.corr_case(300 + b, n, T_gt, 0.3, noise=0.03)
        src.append(torch.from_numpy(p0).cuda()); tgt.append(torch.from_numpy(p1).cuda())
    out = m.SC2_PCR_batch(src, tgt)
    got_T = torch.stack([t for t, _ in out]).cpu().numpy()          # ordered on the current stream after the join
    for b, n in enumerate(sizes):
        T1, f1 = m.SC2_PCR(src[b][None], tgt[b][None])
        np.testing.assert_array_equal(got_T[b], T1[0].cpu().numpy())
        np.testing.assert_array_equal(out[b][1].cpu().numpy(), f1[0].cpu().numpy())
        if n >= 800:
            np.testing.assert_allclose(got_T[b][:3, :3], T_gt[:3, :3], atol=0.02)
    # throughput sanity: the batch must not be slower than the per-pair loop
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        m.SC2_PCR_batch(src, tgt)
    torch.cuda.synchronize(); tb = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(3):
        for b in range(len(sizes)):
            m.SC2_PCR(src[b][None], tgt[b][None])
    torch.cuda.synchronize(); tl = time.perf_counter() - t0
    print(f"SC2-PCR {len(sizes)} pairs: batched {tb / 3 * 1e3:.2f} ms, loop {tl / 3 * 1e3:.2f} ms")
    assert tb < tl * 1.1


def test_seed_stage_paths_agree_bit_for_bit():
    """The per-seed stage has two ways to get a seed's second-order counts (one wave per candidate; dense blocks of 64 seeds with
    the rows in registers) and two ways to pick the top-k1 (short list; histogram): every combination must give the same poses and
    the same seed-wise fitness, bit for bit - on noisy pairs at three inlier ratios, on a pair whose inliers are EXACT (thousands
    of equal counts: the short list overflows by itself) and on ragged sizes.  Ref: SC2_PCR.py:61-108 (cal_seed_trans)."""
    import eyoc_amd
    from eyoc_amd import _lib as L
    m = eyoc_amd.Matcher(inlier_threshold=0.6, d_thre=0.1, ratio=0.2, nms_radius=0.6, max_points=8000, k1=30, k2=20, num_iterations=20)
    T = gi.rigid(0.02, -0.01, 0.1, 4.0, 0.3, -0.2)
    cases = [(700, 3000, 0.1, 0.03), (701, 5000, 0.3, 0.03), (702, 8000, 0.6, 0.05), (703, 4000, 0.6, 0.0), (704, 2500, 1.0, 0.0),
             (705, 777, 0.4, 0.02)]
    src, tgt = [], []
    for seed, n, frac, noise in cases:
        p0, p1, _ = gi.corr_case(seed, n, T, frac, noise=noise)
        src.append(torch.from_numpy(p0).cuda()); tgt.append(torch.from_numpy(p1).cuda())
    lib, ctx = L.load(), L.ctx(0)
    out = {}
    try:
        for cap in (1024, 0, 40):
            for x in (2, 0, 6, -1):
                lib.eyoc_sc2pcr_set_shortlist_cap(ctx, cap)
                lib.eyoc_sc2pcr_set_dense_threshold(ctx, x)
                out[cap, x] = [(Tb.cpu().numpy(), fb.cpu().numpy()) for Tb, fb in m.SC2_PCR_batch(src, tgt)]
    finally:
        lib.eyoc_sc2pcr_set_shortlist_cap(ctx, 1024)
        lib.eyoc_sc2pcr_set_dense_threshold(ctx, 0)
    ref = out[0, -1]                        # histogram selection, no dense blocks: the round-4 algorithm
    for b, (seed, n, frac, noise) in enumerate(cases):
        assert np.isfinite(ref[b][0]).all()
        if frac >= 0.3:
            np.testing.assert_allclose(ref[b][0], T, atol=0.05)
    for key, res in out.items():
        for b in range(len(cases)):
            np.testing.assert_array_equal(res[b][0], ref[b][0], err_msg=f"pose, case {b}, (cap, x) = {key}")
            np.testing.assert_array_equal(res[b][1], ref[b][1], err_msg=f"fitness, case {b}, (cap, x) = {key}")


def test_round6_kernels_equal_their_round5_forms_bit_for_bit():
    """Round 6 rewrote four kernels of the back-end without touching a result (csrc/sc2pcr.hip): the CSR fill compacts a row before it
    evaluates the cross lengths; the mask kernel decides with v_sqrt_f32 and a band and evaluates the correctly rounded expression
    only for undecided lanes; the NMS and seed-fitness sweeps compare the SQUARED length with T(r) = min {x : sqrtf(x) >= r}; the
    seeds' 3 x 3 Kabsch solves and inlier counts run lane-per-seed behind the wave-per-seed kernel.
    ``eyoc_sc2pcr_select_kernels`` brings the round-5 forms back one by one: poses and all seed-wise fitness values must not move by a
    bit - noisy pairs at three inlier ratios, exact inliers (cross lengths of exactly 0 and thresholds met from both sides), ragged and
    tiny sizes, a second parameter set (3DMatch-like thresholds), and coordinates scaled to 1e4 m (large roots: wide bands)."""
    import eyoc_amd
    from eyoc_amd import _lib as L
    T = gi.rigid(0.02, -0.01, 0.1, 4.0, 0.3, -0.2)
    cases = [(710, 3000, 0.1, 0.03, 1.0), (711, 5000, 0.3, 0.03, 1.0), (712, 8000, 0.6, 0.05, 1.0), (713, 4000, 0.6, 0.0, 1.0),
             (714, 777, 0.4, 0.02, 1.0), (715, 65, 0.5, 0.01, 1.0), (716, 2000, 0.3, 0.03, 1.0e4), (717, 4097, 0.2, 0.1, 1.0)]
    src, tgt = [], []
    for seed, n, frac, noise, scale in cases:
        p0, p1, _ = gi.corr_case(seed, n, T, frac, noise=noise)
        src.append(torch.from_numpy(p0 * np.float32(scale)).cuda()); tgt.append(torch.from_numpy(p1 * np.float32(scale)).cuda())
    lib, ctx = L.load(), L.ctx(0)
    matchers = [eyoc_amd.Matcher(inlier_threshold=0.6, d_thre=0.1, ratio=0.2, nms_radius=0.6, max_points=8200, k1=30, k2=20, num_iterations=20),
                eyoc_amd.Matcher(inlier_threshold=0.10, d_thre=0.1, ratio=0.1, nms_radius=0.10, max_points=8200, k1=30, k2=20, num_iterations=10)]
    prev = lib.eyoc_sc2pcr_select_kernels(ctx, 0)
    assert prev == 0
    try:
        for mi, m in enumerate(matchers):
            out = {}
            for bits in (15, 0, 1, 2, 4, 8):
                assert lib.eyoc_sc2pcr_select_kernels(ctx, bits) >= 0
                out[bits] = [(Tb.cpu().numpy(), fb.cpu().numpy()) for Tb, fb in m.SC2_PCR_batch(src, tgt)]
            for bits in (0, 1, 2, 4, 8):
                for b in range(len(cases)):
                    np.testing.assert_array_equal(out[bits][b][0], out[15][b][0], err_msg=f"pose, matcher {mi}, case {b}, legacy bits {bits}")
                    np.testing.assert_array_equal(out[bits][b][1], out[15][b][1], err_msg=f"fitness, matcher {mi}, case {b}, legacy bits {bits}")
            if mi == 0:
                for b in (1, 2, 3):
                    np.testing.assert_allclose(out[0][b][0], T, atol=0.05)
    finally:
        lib.eyoc_sc2pcr_select_kernels(ctx, 0)


def test_round6_kernels_at_the_largest_problem_size():
    """n = 16384 (MAX_N: 256 mask words per row - four chunks of the compacted CSR fill, the 64-word-per-wave dense count kernel, 64 KB
    count rows) and n = 8193 (129 words: the first size of that kernel), round-6 kernels against their round-5 forms bit for bit, and
    the pose recovered."""
    import eyoc_amd
    from eyoc_amd import _lib as L
    T = gi.rigid(0.01, 0.02, -0.08, -3.0, 1.3, 0.1)
    lib, ctx = L.load(), L.ctx(0)
    m = eyoc_amd.Matcher(inlier_threshold=0.6, d_thre=0.1, ratio=0.2, nms_radius=0.6, max_points=16384, k1=30, k2=20, num_iterations=20)
    try:
        for seed, n in ((720, 16384), (721, 8193)):
            p0, p1, _ = gi.corr_case(seed, n, T, 0.15, noise=0.03)
            src, tgt = torch.from_numpy(p0).cuda()[None], torch.from_numpy(p1).cuda()[None]
            out = {}
            for bits in (15, 0):
                lib.eyoc_sc2pcr_select_kernels(ctx, bits)
                Tb, fb = m.SC2_PCR(src, tgt)
                out[bits] = (Tb[0].cpu().numpy(), fb[0].cpu().numpy())
            np.testing.assert_array_equal(out[0][0], out[15][0], err_msg=f"pose, n = {n}")
            np.testing.assert_array_equal(out[0][1], out[15][1], err_msg=f"fitness, n = {n}")
            np.testing.assert_allclose(out[0][0], T, atol=0.05)
            assert out[0][1].shape == (int(n * 0.2),) and out[0][1].max() > 0.1 * n
    finally:
        lib.eyoc_sc2pcr_select_kernels(ctx, 0)


def test_harness_sc2pcr_path_equals_per_pair_estimator():
    """RegistrationPipeline with use_RANSAC=False (scripts/test_kitti.py:179-181) batches the matching and the
    SC2-PCR of all pairs; the poses are bit-identical to looping ``Matcher.estimator`` with the same draws."""
    import eyoc_amd
    from eyoc_amd import synthetic as syn
    from eyoc_amd.harness import DeviceBatch, RegistrationConfig, RegistrationPipeline
    model = eyoc_amd.load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, conv1_kernel_size=5, normalize_feature=True)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_weights().items()})
    model = model.cuda().eval()
    cfg = RegistrationConfig(use_RANSAC=False)
    pipe = RegistrationPipeline(model, cfg)
    pairs, seeds = [syn.make_pair(s) for s in range(3)], [0, 1, 2]
    batch = DeviceBatch(pairs, seeds, torch.device("cuda"), cfg.n_points)
    T = pipe.register(batch, seed=5, return_device=True).cpu().numpy()
    # the same thing pair by pair
    F = pipe.features(batch).F
    F0, F1 = F.index_select(0, batch.sel0), F.index_select(0, batch.sel1)
    n = batch.n_points
    rng = np.random.RandomState(5)
    for p in range(3):
        Tp, _, _, _, _ = pipe.matcher.estimator(batch.xyz0[p][None], batch.xyz1[p][None], F0[p * n:(p + 1) * n][None],
                                                F1[p * n:(p + 1) * n][None], rng=rng)
        np.testing.assert_array_equal(T[p], Tp[0].cpu().numpy())
