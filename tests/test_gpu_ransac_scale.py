"""RANSAC at the reference's production settings (scripts/test_kitti.py:169-177: 4 000 000 hypotheses, confidence clamped to
1 = no early exit) and through the paths only large runs reach: the per-pair hypothesis indexing at H = 4 M, more survivors
than the 2^20-entry transform store (k_count_overflow), launch chunks smaller than the batch (caller-owned workspace), and the
context-owned scratch entry points.  Checker: oracle/ransac.py with the shared counter sampler."""
import ctypes as C

import numpy as np
import pytest
import torch

import _inputs as gi

pytestmark = pytest.mark.gpu


def _lib():
    from eyoc_amd import _lib as L
    return L, L.load()


def _same(res, ref):
    assert res.survivors == ref["survivors"]
    assert res.best_hypothesis == ref["best_h"]
    assert res.inliers == ref["inliers"]
    assert res.inlier_rmse == pytest.approx(ref["rmse"], rel=1e-5)
    np.testing.assert_allclose(res.transformation, ref["T"], atol=1e-5)


def test_ransac_matches_oracle_at_four_million_hypotheses():
    """n = 5000 correspondences at an inlier ratio of 0.1: ~400 of the 4 M hypotheses survive; the winner sits at h > 2 M."""
    import eyoc_amd
    from oracle import ransac as orn
    T = gi.rigid(0.01, -0.02, 0.15, 9.0, 0.5, 0.1)
    p0, p1, _ = gi.corr_case(91, 5000, T, 0.1, noise=0.03)
    res = eyoc_amd.ransac_from_correspondences(torch.from_numpy(p0), torch.from_numpy(p1), torch.arange(5000), 0.3, 4000000, seed=11)
    ref = orn.ransac(p0, p1, np.arange(5000), 0.3, 4000000, seed=11)
    print(f"4 M hypotheses: survivors {res.survivors}, inliers {res.inliers}, best h {res.best_hypothesis}")
    assert ref["best_h"] > 2000000 and ref["survivors"] > 100
    _same(res, ref)
    np.testing.assert_allclose(res.transformation[:3, :3], T[:3, :3], atol=0.03)


def test_more_survivors_than_the_transform_store_holds():
    """64 exact-ish inliers, 1.2 M hypotheses: 95 % of them survive - 1.14 M > 2^20 - so ~95 000 survivors (which ones depends
    on the order the survivor list filled in) are scored by k_count_overflow and, at the largest count, by k_rmse's re-derivation."""
    import eyoc_amd
    from oracle import ransac as orn
    T = gi.rigid(0.01, -0.02, 0.15, 9.0, 0.5, 0.1)
    p0, p1, _ = gi.corr_case(300, 64, T, 1.0, noise=0.02)
    H = 1200000
    res = eyoc_amd.ransac_from_correspondences(torch.from_numpy(p0), torch.from_numpy(p1), torch.arange(64), 0.05, H, seed=1)
    ref = orn.ransac(p0, p1, np.arange(64), 0.05, H, seed=1)
    print(f"overflow case: survivors {res.survivors} (store 2^20), inliers {res.inliers}, best h {res.best_hypothesis}")
    assert ref["survivors"] > (1 << 20)
    _same(res, ref)


@pytest.mark.parametrize("n,frac,H,seed", [(1500, 0.7, 60000, 2), (777, 0.5, 20000, 3)])
def test_overflow_scorer_produces_the_winning_count(n, frac, H, seed):
    """The transform store shrunk to 8 entries (eyoc_ransac_transform_store): all but 8 of the thousands of survivors - the
    winner among them - get their count from k_count_overflow and their RMSE from k_rmse's re-derived transform.  Same answer
    as the oracle and, bit for bit, as the stored-transform path."""
    import eyoc_amd
    from oracle import ransac as orn
    L, lib = _lib()
    p0, p1, _ = gi.corr_case(40 + n, n, gi.rigid(0.2, -0.1, 0.3, 1.0, -2.0, 0.5), frac)
    p1 = (p1 + np.random.default_rng(n).normal(0, 0.03, p1.shape)).astype(np.float32)
    args = (torch.from_numpy(p0), torch.from_numpy(p1), torch.arange(n), 0.3, H)
    full = eyoc_amd.ransac_from_correspondences(*args, seed=seed)
    prev = L.knob("eyoc_ransac_transform_store", 8)
    try:
        assert L.knob("eyoc_ransac_transform_store", -1) == 8
        res = eyoc_amd.ransac_from_correspondences(*args, seed=seed)
    finally:
        L.knob("eyoc_ransac_transform_store", prev)
    ref = orn.ransac(p0, p1, np.arange(n), 0.3, H, seed=seed)
    assert res.survivors > 1000
    _same(res, ref)
    assert (res.survivors, res.best_hypothesis, res.inliers, res.inlier_rmse) == (full.survivors, full.best_hypothesis, full.inliers, full.inlier_rmse)
    np.testing.assert_array_equal(res.transformation, full.transformation)


def _ragged_batch():
    T = gi.rigid(0.02, 0.01, -0.12, 5.0, -0.3, 0.2)
    sizes = [400, 1500, 7000, 5, 900, 2048, 33, 640, 1200, 777, 3100]
    src, tgt, corr, seg_s, seg_t = [], [], [], [0], [0]
    for b, n in enumerate(sizes):
        p0, p1, _ = gi.corr_case(200 + b, n, T, 0.25 if n > 100 else 1.0, noise=0.03)
        perm = np.random.default_rng(b).permutation(n)
        src.append(p0); tgt.append(p1[perm]); corr.append(np.argsort(perm))
        seg_s.append(seg_s[-1] + n); seg_t.append(seg_t[-1] + n)
    dev = torch.device("cuda")
    return (torch.from_numpy(np.concatenate(src)).to(dev), torch.from_numpy(np.concatenate(tgt)).to(dev),
            torch.from_numpy(np.concatenate(corr)).to(dev), seg_s, seg_t)


def test_results_do_not_depend_on_the_launch_chunk_or_on_who_owns_the_scratch():
    """11 ragged pairs: one launch chunk (default budget), chunks of 5 / 2 / 1 pairs (workspace budgets: the chunk is the batch, halved until it fits), the context-owned
    scratch entry point (eyoc_ransac_batched: chunk sized from hipMemGetInfo) - identical result records; a workspace too
    small for a single pair is refused with EYOC_ERR_WORKSPACE, nothing allocated, nothing written."""
    from eyoc_amd import registration as reg
    L, lib = _lib()
    s, t, c, seg_s, seg_t = _ragged_batch()
    P, H = len(seg_s) - 1, 60000
    want = reg.ransac_batched_from_correspondences(s, t, c, seg_s, seg_t, 0.3, H, seed=40).cpu().numpy()
    sizes = {k: int(lib.eyoc_ransac_workspace_bytes(L.ctx(0), k, seg_s[-1], H, 0)) for k in (1, 2, 5, P)}
    assert sizes[1] < sizes[2] < sizes[5] < sizes[P]
    for k in (5, 2, 1):
        assert int(lib.eyoc_ransac_workspace_bytes(L.ctx(0), P, seg_s[-1], H, sizes[k])) == sizes[k]       # the largest chunk within the budget
        got = reg.ransac_batched_from_correspondences(s, t, c, seg_s, seg_t, 0.3, H, seed=40, workspace_budget=sizes[k]).cpu().numpy()
        np.testing.assert_array_equal(got, want, err_msg=f"chunk of {k} pairs")
    # context-owned scratch
    ss, st = (C.c_int32 * (P + 1))(*seg_s), (C.c_int32 * (P + 1))(*seg_t)
    p = L.RansacParams(0.3, 0.9, H, 40)
    res = torch.zeros((P, C.sizeof(L.RansacResult)), dtype=torch.uint8, device="cuda")
    L.check(lib.eyoc_ransac_batched(L.ctx(), L.ptr(s), L.ptr(t), L.ptr(c), ss, st, P, C.byref(p), L.ptr(res), L.stream_ptr()))
    np.testing.assert_array_equal(res.cpu().numpy(), want)
    one = torch.zeros(C.sizeof(L.RansacResult), dtype=torch.uint8, device="cuda")
    p1 = L.RansacParams(0.3, 0.9, H, 41)
    L.check(lib.eyoc_ransac(L.ctx(), L.ptr(s[seg_s[1]:]), L.ptr(t[seg_t[1]:]), L.ptr(c[seg_s[1]:]), seg_s[2] - seg_s[1], C.byref(p1),
                            L.ptr(one), L.stream_ptr()))
    np.testing.assert_array_equal(one.cpu().numpy(), want[1])
    # too small
    ws = L.workspace(sizes[1] - 256, s.device)
    res.fill_(7)
    rc = lib.eyoc_ransac_batched_ws(L.ctx(), L.ptr(s), L.ptr(t), L.ptr(c), ss, st, P, C.byref(p), L.ptr(res), L.ptr(ws), ws.numel(),
                                    L.stream_ptr())
    assert rc == -3 and b"workspace" in lib.eyoc_last_error()
    torch.cuda.synchronize()
    assert int(res.min()) == 7


@pytest.mark.parametrize("shift,scale,noise", [(0.0, 1.0, 0.1), (3.0e4, 1.0, 0.1), (0.0, 1.0e5, 0.0), (0.0, 1.0, 0.29)])
def test_packed_fp32_count_decides_like_the_fp64_count(shift, scale, noise):
    """k_count sweeps `r_ref + dR p + dt` in packed fp32 against `thr^2 -+ band` and recounts what falls inside the band in fp64 (round
    5).  Its counts must be the fp64 sweep's (k_count_fp64: pruning off) and the oracle's where the band is (a) narrow (the bench's
    regime), (b) wide because the clouds sit 30 km from the origin (fp32 coordinates resolve 2 mm there: many residuals undecided),
    (c) unusable - coordinates of 1e7, the band exceeds thr^2, every block counts in fp64 - and (d) with the inliers' residuals spread
    right up to the threshold (noise ~ max_distance: the densest band)."""
    import eyoc_amd
    from oracle import ransac as orn
    L, lib = _lib()
    n, H, thr = 3000, 200000, 0.3
    T = gi.rigid(0.02, -0.01, 0.1, 2.0, -1.0, 0.3)
    p0, p1, _ = gi.corr_case(500 + int(noise * 100), n, T, 0.5, noise=noise)
    if scale != 1.0:
        p0, p1, thr = (p0 * scale).astype(np.float32), (p1 * scale).astype(np.float32), thr * scale
    if shift:
        p0, p1 = (p0 + np.float32(shift)).astype(np.float32), (p1 + np.float32(shift)).astype(np.float32)
    args = (torch.from_numpy(p0), torch.from_numpy(p1), torch.arange(n), thr, H)
    packed = eyoc_amd.ransac_from_correspondences(*args, seed=5)
    prev = L.knob("eyoc_ransac_select_pruning", 0)
    try:
        plain = eyoc_amd.ransac_from_correspondences(*args, seed=5)
    finally:
        L.knob("eyoc_ransac_select_pruning", prev)
    ref = orn.ransac(p0, p1, np.arange(n), thr, H, seed=5)
    print(f"shift {shift} scale {scale} noise {noise}: survivors {packed.survivors}, inliers {packed.inliers} (oracle {ref['inliers']})")
    assert packed.survivors > 50
    assert (packed.survivors, packed.best_hypothesis, packed.inliers, packed.inlier_rmse) == \
           (plain.survivors, plain.best_hypothesis, plain.inliers, plain.inlier_rmse)
    np.testing.assert_array_equal(packed.transformation, plain.transformation)
    assert (packed.survivors, packed.best_hypothesis, packed.inliers) == (ref["survivors"], ref["best_h"], ref["inliers"])


def test_count_bound_never_changes_the_winner():
    """Round 6 (eyoc_ransac_select_pruning 2, the default): in k_count a survivor stops counting once its count so far plus the records
    still ahead of it is below the largest count any survivor of the pair has reached.  The records a caller sees - survivors, winning
    hypothesis, its inlier count, RMSE and transform - must be those of the reference-pruned count (1) and of the full fp64 sweeps (0),
    byte for byte: pairs at inlier ratios 0.15 - 0.7 (hundreds to 100 k survivors), EXACT inliers (thousands of survivors tie at the
    largest count: RMSE and hypothesis number decide), a pair with no survivor, ragged sizes, and the same batch in two launch chunks."""
    import eyoc_amd
    from eyoc_amd import registration as reg
    L, lib = _lib()
    T = gi.rigid(0.03, -0.02, 0.15, 3.0, -1.5, 0.4)
    cases = [(900, 5000, 0.3, 0.05), (901, 5000, 0.15, 0.05), (902, 4000, 0.7, 0.03), (903, 3000, 0.5, 0.0), (904, 1500, 0.02, 0.05),
             (905, 777, 0.4, 0.1), (906, 5000, 0.45, 0.12)]
    src, tgt, seg = [], [], [0]
    for seed, n, frac, noise in cases:
        p0, p1, _ = gi.corr_case(seed, n, T, frac, noise=noise)
        src.append(p0); tgt.append(p1); seg.append(seg[-1] + n)
    # the bound is only switched on for launch chunks of >= 32 pairs (below that its polling of one word per pair costs more than it
    # saves): the seven cases are repeated to 35 pairs - pair b uses seed + b, so the copies draw different hypotheses
    reps = 5
    cases = cases * reps
    src, tgt = src * reps, tgt * reps
    seg = [0]
    for _, n, _, _ in cases:
        seg.append(seg[-1] + n)
    s, t = torch.from_numpy(np.concatenate(src)), torch.from_numpy(np.concatenate(tgt))
    corr = torch.cat([torch.arange(n) for _, n, _, _ in cases])
    out = {}
    prev = L.knob("eyoc_ransac_select_pruning", -7)
    assert prev == 2
    try:
        for mode in (2, 1, 0):
            L.knob("eyoc_ransac_select_pruning", mode)
            for budget in (None, 1 << 28):                 # one launch chunk of 35 pairs (bound on in mode 2) / several smaller ones (bound off)
                out[mode, budget] = reg.ransac_batched_from_correspondences(s, t, corr, seg, seg, 0.3, 400000, seed=11, workspace_budget=budget).cpu().numpy()
    finally:
        L.knob("eyoc_ransac_select_pruning", prev)
    ref = out[0, None]
    for key, res in out.items():
        np.testing.assert_array_equal(res, ref, err_msg=f"(mode, budget) = {key}")
    dec = [reg.decode_ransac_result(torch.from_numpy(ref[b]), cases[b][1]) for b in range(len(cases))]
    print("survivors", [d.survivors for d in dec], "inliers", [d.inliers for d in dec])
    assert dec[0].survivors > 1000 and dec[2].survivors > 20000 and dec[3].inliers >= int(0.45 * 3000)
    for b in (0, 2, 3, 6):
        np.testing.assert_allclose(dec[b].transformation, T, atol=0.05)
