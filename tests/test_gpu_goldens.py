"""The product's glue functions against golden vectors from the REFERENCE's own code (tests/golden/make_golden.py, G7-G9:
``lib.trainer`` and ``scripts.test_kitti`` imported with a codec alias and empty stand-ins for the absent libraries, no
arithmetic stubbed): the hardest-contrastive loss and its gradients, ratio test + top-k, find_corr, random_sample, the NN
distance metric.  Same seeded inputs, same global ``np.random`` seeding as the generator."""
import json
import os

import numpy as np
import pytest
import torch

import _inputs as gi

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("i", [0, 1, 2])
def test_loss_and_gradients_match_the_reference(golden_dir, i):
    """eyoc_amd.autograd.contrastive_hardest_negative_loss (lib/trainer.py:935-991): the default ``rng`` IS the global
    ``np.random``, so seeding it like the generator did reproduces the reference's three draws."""
    from eyoc_amd.autograd import contrastive_hardest_negative_loss
    g = _load(golden_dir, "g7_loss.npz")
    seed, n0, n1, npairs, num_pos, nhn = json.loads(str(g["cases"]))[i]
    F0n, F1n, pairs = gi.loss_case(seed, n0, n1, npairs)
    F0, F1 = torch.from_numpy(F0n).cuda().requires_grad_(True), torch.from_numpy(F1n).cuda().requires_grad_(True)
    np.random.seed(seed)
    pos, neg = contrastive_hardest_negative_loss(F0, F1, torch.from_numpy(pairs), num_pos=num_pos, num_hn_samples=nhn)
    (pos + neg).backward()
    np.testing.assert_allclose(float(pos.detach()), float(g[f"pos{i}"]), rtol=1e-5)
    np.testing.assert_allclose(float(neg.detach()), float(g[f"neg{i}"]), rtol=1e-5, equal_nan=True)
    for got, want in ((F0.grad.cpu().numpy(), g[f"gF0_{i}"]), (F1.grad.cpu().numpy(), g[f"gF1_{i}"])):
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-4 * np.abs(want).max())
    print(f"loss case {i}: pos {float(pos.detach()):.6f} neg {float(neg.detach()):.6f}; max grad err "
          f"{np.abs(F0.grad.cpu().numpy() - g[f'gF0_{i}']).max() / np.abs(g[f'gF0_{i}']).max():.2e}")


@pytest.mark.parametrize("i", [0, 1, 2])
def test_ratio_test_and_topk_match_the_reference(golden_dir, i):
    """eyoc_knn2 + eyoc_lowe_topk against calculate_ratio_test / get_topk_matches (lib/trainer.py:993-1016) on the
    cosines ``1 - 0.5 d`` of :1066-1070: neighbour, both distances and the weights bit-exact, the top-k order exact wherever
    the weight is unique (``torch.topk`` leaves ties open)."""
    from eyoc_amd import labels
    g = _load(golden_dir, "g8_labels.npz")
    seed, n0, n1, k = json.loads(str(g["cases"]))[i]
    F0, F1 = gi.nn_case(seed, n0, n1)
    idx, d1, d2 = labels.knn2_segmented(torch.from_numpy(F0).cuda(), torch.from_numpy(F1).cuda(), [0, n0], [0, n1])
    np.testing.assert_array_equal(idx.cpu().numpy(), g[f"idx{i}"])
    np.testing.assert_array_equal(d1.cpu().numpy(), g[f"d1_{i}"])
    np.testing.assert_array_equal(d2.cpu().numpy(), g[f"d2_{i}"])
    src, w = labels.lowe_topk(d1, d2, k)
    src, w = src.cpu().numpy(), w.cpu().numpy()
    top = g[f"top{i}"]
    assert len(src) == min(k, n0)
    np.testing.assert_array_equal(w, top)
    vals, counts = np.unique(top, return_counts=True)
    uniq = counts[np.searchsorted(vals, top)] == 1
    np.testing.assert_array_equal(src[uniq], g[f"src{i}"][uniq])
    np.testing.assert_array_equal(idx.cpu().numpy()[src][uniq], g[f"tgt{i}"][uniq])


@pytest.mark.parametrize("i", [0, 1, 2, 3])
def test_match_and_filter_corr_matches_the_reference(golden_dir, i):
    """eyoc_amd.labels.match_and_filter_corr (eyoc_knn2, eyoc_lowe_topk, eyoc_pair_filter / _similarity and the re-collation) against
    the reference's own method (G10, lib/trainer.py:1025-1151 run unmodified on three pairs with stored K = 2 neighbours): collated
    matches and per-pair filtered survivors, up to the order ``torch.topk`` leaves open between equal weights."""
    from eyoc_amd import labels
    from test_oracle_golden import same_rows_up_to_topk_ties
    g = _load(golden_dir, "g10_match_filter.npz")
    ff, sf, fd = json.loads(str(g["cases"]))[i]
    C0s, F0s, C1s, F1s = gi.label_batch_case(101)
    t = lambda xs: [torch.from_numpy(x) for x in xs]
    m, unc = labels.match_and_filter_corr(t(C0s), t(F0s), t(C1s), t(F1s), radius=20, feature_filter=ff, spatial_filter=sf, frame_distance=fd,
                                          dist_sim_map=gi.dist_sim_table(), similarity_thresh=0.3)
    same_rows_up_to_topk_ties(m.numpy(), g[f"matches{i}"], "matches")
    for p, u in enumerate(unc):
        same_rows_up_to_topk_ties(u.cpu().numpy(), g[f"unc{i}_{p}"], f"pair {p}")


@pytest.mark.parametrize("i", [0, 1, 2])
def test_find_corr_matches_the_reference(golden_dir, i):
    """scripts/test_kitti.py:28-42: drawn rows exact, neighbours exact up to fp32 near-ties of the distance."""
    import eyoc_amd
    from oracle import matching as om
    g = _load(golden_dir, "g9_eval.npz")
    seed, n0, n1, sub = json.loads(str(g["cases"]))[i]
    F0, F1 = gi.nn_case(seed, n0, n1)
    x0 = np.zeros((n0, 3), np.float32); x0[:, 0] = np.arange(n0)
    x1 = np.zeros((n1, 3), np.float32); x1[:, 0] = np.arange(n1)
    np.random.seed(seed)
    a, b = eyoc_amd.find_corr(torch.from_numpy(x0), torch.from_numpy(x1), torch.from_numpy(F0).cuda(), torch.from_numpy(F1).cuda(),
                              subsample_size=sub)
    assert eyoc_amd.find_correspondences is eyoc_amd.find_corr
    a, b = a[:, 0].numpy().astype(np.int32), b[:, 0].numpy().astype(np.int32)
    np.testing.assert_array_equal(a, g[f"corr0_{i}"])
    diff = np.nonzero(b != g[f"corr1_{i}"])[0]
    assert len(diff) <= max(1, len(b) // 1000)
    for r in diff:
        D = om.sqdist_rows(F0[a[r]:a[r] + 1], F1)[0]
        assert abs(D[b[r]] - D[g[f"corr1_{i}"][r]]) <= 4e-6


def test_random_sample_and_metrics_match_the_reference(golden_dir):
    import eyoc_amd
    from eyoc_amd import metrics
    g = _load(golden_dir, "g9_eval.npz")
    for j, (n, N) in enumerate(((1000, 300), (200, 500), (64, 64))):
        pts = np.zeros((n, 3), np.float32); pts[:, 0] = np.arange(n)
        feats = torch.arange(n * 4, dtype=torch.float32).reshape(n, 4).cuda()
        np.random.seed(100 + j)
        p, f = eyoc_amd.random_sample(pts, feats, N)
        np.testing.assert_array_equal(np.asarray(p)[:, 0].astype(np.int32), g[f"rs_p{j}"])
        np.testing.assert_array_equal(f.cpu().numpy(), g[f"rs_f{j}"])
    T = g["T"]
    pts = ((gi._u(95, 500, 3) - 0.5) * 60).astype(np.float32)
    tgt = (pts @ T[:3, :3].T + T[:3, 3] + 0.05 * gi._normal(96, 500, 3)).astype(np.float32)
    np.testing.assert_allclose(metrics.apply_transform(pts, T), g["applied"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(metrics.evaluate_nn_dist(pts, tgt, T), g["nn_dist"], rtol=2e-5, atol=2e-6)
