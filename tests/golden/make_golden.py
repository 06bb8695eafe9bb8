"""Generate the golden OUTPUT vectors by running the reference's own importable functions.

Run in the build container only (``python tests/golden/make_golden.py``): it imports
``/root/reference`` read-only.  MinkowskiEngine / open3d are imported by some reference modules at
module top but not used by the functions exercised here, so empty ``sys.modules`` stubs stand in for
them.  Nothing of the reference is copied: the fixtures hold inputs' seeds and the
reference's numeric outputs only.

  g1_nn.npz      lib.eval.find_nn_gpu / lib.metrics.pdist
  g2_irls.npz    util.transform_estimation.est_quad_linear_robust
  g3_kabsch.npz  scripts.SC2_PCR.common.rigid_transform_3d
  g4_sc2pcr.npz  scripts.SC2_PCR.SC2_PCR.Matcher.SC2_PCR / cal_leading_eigenvector (KITTI config)
  g5_se3.npz     scripts.SC2_PCR.utils.SE3.transform / integrate_trans
  g6_match.npz   scripts.SC2_PCR.SC2_PCR.Matcher.match_pair (the method hard-codes one ``.cuda()`` on an index tensor,
                 SC2_PCR.py:299; it runs here with ``torch.Tensor.cuda`` patched to the identity - nothing else changes)
  g7_loss.npz    lib.trainer.CorrespondenceExtensionTrainer.contrastive_hardest_negative_loss (lib/trainer.py:935-991): both
                 loss terms AND d(pos + neg)/dF0, /dF1, global ``np.random`` seeded per case
  g8_labels.npz  lib.trainer.CorrespondenceExtensionTrainer.calculate_ratio_test / get_topk_matches (lib/trainer.py:993-1016)
  g10_match_filter.npz  lib.trainer.CorrespondenceExtensionTrainer.match_and_filter_corr (lib/trainer.py:1025-1151) with STORED K = 2 neighbours
  g9_eval.npz    scripts.test_kitti.find_corr / random_sample / apply_transform / evaluate_nn_dist (scripts/test_kitti.py:28-73)
``lib.trainer`` and ``scripts.test_kitti`` import through ``_refimport.install()`` (codec alias + EMPTY stand-ins for the
absent libraries; no arithmetic is stubbed).
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
sys.path.insert(0, REF)
import _refimport  # noqa: E402
_refimport.install()

import _inputs as gi  # noqa: E402
from lib.eval import find_nn_gpu  # noqa: E402
from lib.metrics import pdist  # noqa: E402
from util.transform_estimation import est_quad_linear_robust  # noqa: E402
from scripts.SC2_PCR.common import rigid_transform_3d  # noqa: E402
from scripts.SC2_PCR.SC2_PCR import Matcher  # noqa: E402
from scripts.SC2_PCR.utils.SE3 import transform, integrate_trans  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def g1():
    out = {}
    for tag, (seed, n0, n1) in {"big": (11, 5000, 5000), "odd": (12, 257, 129), "wide": (13, 300, 4000)}.items():
        F0, F1 = gi.nn_case(seed, n0, n1)
        t0, t1 = torch.from_numpy(F0), torch.from_numpy(F1)
        inds, d = find_nn_gpu(t0, t1, nn_max_n=500, return_distance=True)
        inds2, d2 = find_nn_gpu(t0, t1, nn_max_n=-1, return_distance=True)
        assert torch.equal(inds, inds2), "nn_max_n must not change the result"
        out[f"{tag}_meta"] = np.array([seed, n0, n1])
        out[f"{tag}_inds"] = inds.numpy().astype(np.int32)
        out[f"{tag}_d2"] = d.numpy()[:, 0]
        indsL, dL = find_nn_gpu(t0, t1, nn_max_n=500, return_distance=True, dist_type="L2")
        out[f"{tag}_inds_l2"] = indsL.numpy().astype(np.int32)
        out[f"{tag}_d_l2"] = dL.numpy()[:, 0]
    A, B = gi.nn_case(14, 16, 8)
    out["small_pdist_sq"] = pdist(torch.from_numpy(A), torch.from_numpy(B), "SquareL2").numpy()
    out["small_pdist_l2"] = pdist(torch.from_numpy(A), torch.from_numpy(B), "L2").numpy()
    np.savez_compressed(os.path.join(HERE, "g1_nn.npz"), **out)


IRLS_CASES = [  # (seed, n, inlier_frac, use_weight, T params)
    (21, 5000, 1.0, False, (0.02, -0.03, 0.30, 2.0, -0.5, 0.1)),
    (22, 5000, 0.7, False, (0.01, 0.02, -0.10, 1.0, 0.3, -0.05)),
    (23, 5000, 0.4, False, (-0.02, 0.01, 0.05, 0.5, 0.2, 0.02)),
    (24, 3000, 0.7, True, (0.03, 0.00, 0.15, -1.2, 0.4, 0.0)),
    (25, 777, 0.9, True, (0.0, 0.0, 0.0, 0.0, 0.0, 0.0)),
]


def g2():
    out = {"cases": np.array(json.dumps(IRLS_CASES))}
    for i, (seed, n, frac, use_w, tp) in enumerate(IRLS_CASES):
        p0, p1, _ = gi.corr_case(seed, n, gi.rigid(*tp), frac)
        w = None
        if use_w:
            w = torch.from_numpy((0.05 + 0.95 * gi._u(seed + 9, n, 1)).astype(np.float32))
        T = est_quad_linear_robust(torch.from_numpy(p0), torch.from_numpy(p1), w)
        out[f"T{i}"] = T.numpy()
    np.savez_compressed(os.path.join(HERE, "g2_irls.npz"), **out)


KABSCH_CASES = [  # (seed, bs, n, kind)
    (31, 1, 5000, "weighted"), (32, 1000, 20, "weighted"), (33, 64, 20, "zeros"),
    (34, 8, 50, "none"), (35, 4, 30, "reflect"), (36, 16, 3, "weighted"),
]


from make_golden_inputs import kabsch_inputs  # noqa: E402


def g3():
    out = {"cases": np.array(json.dumps(KABSCH_CASES))}
    for i, case in enumerate(KABSCH_CASES):
        A, B, w = kabsch_inputs(*case)
        T = rigid_transform_3d(torch.from_numpy(A), torch.from_numpy(B),
                               None if w is None else torch.from_numpy(w.copy()))
        out[f"T{i}"] = T.numpy()
    np.savez_compressed(os.path.join(HERE, "g3_kabsch.npz"), **out)


SC2_CASES = [  # (seed, n, inlier_frac, T params)
    (41, 2000, 0.10, (0.01, -0.02, 0.12, 12.0, 0.4, 0.1)),
    (42, 2000, 0.30, (-0.02, 0.01, -0.08, 7.0, -0.3, 0.0)),
    (43, 2000, 0.60, (0.00, 0.03, 0.17, 18.0, 0.2, -0.1)),
    (44, 500, 0.50, (0.02, 0.00, 0.05, 5.0, 0.0, 0.0)),
]
KITTI_CFG = dict(inlier_threshold=0.6, num_node=8000, use_mutual=False, d_thre=0.1,
                 num_iterations=20, ratio=0.2, nms_radius=0.6, max_points=8000, k1=30, k2=20)


def g4():
    out = {"cases": np.array(json.dumps(SC2_CASES)), "cfg": np.array(json.dumps(KITTI_CFG))}
    m = Matcher(**KITTI_CFG)
    for i, (seed, n, frac, tp) in enumerate(SC2_CASES):
        p0, p1, inl = gi.corr_case(seed, n, gi.rigid(*tp), frac, noise=0.03)
        T, fit = m.SC2_PCR(torch.from_numpy(p0)[None], torch.from_numpy(p1)[None])
        out[f"T{i}"] = T[0].numpy()
        out[f"fitmax{i}"] = np.array(float(fit.max()))
    # leading eigenvector of a fixed symmetric non-negative 256x256 matrix
    M = gi._u(45, 256, 256)
    M = ((M + M.T) * 0.5).astype(np.float32)
    np.fill_diagonal(M, 0)
    out["eig_vec"] = m.cal_leading_eigenvector(torch.from_numpy(M)[None])[0].numpy()
    np.savez_compressed(os.path.join(HERE, "g4_sc2pcr.npz"), **out)


def g5():
    pts = ((gi._u(51, 3, 40, 3) - 0.5) * 10).astype(np.float32)
    R = np.stack([gi.rot_zyx(*((gi._u(52 + b, 3) - 0.5) * 2)) for b in range(3)]).astype(np.float32)
    t = ((gi._u(55, 3, 3, 1) - 0.5) * 5).astype(np.float32)
    T = integrate_trans(torch.from_numpy(R), torch.from_numpy(t))
    out = {"T": T.numpy(), "warped": transform(torch.from_numpy(pts), T).numpy(),
           "warped0": transform(torch.from_numpy(pts[0]), T[0]).numpy()}
    np.savez_compressed(os.path.join(HERE, "g5_se3.npz"), **out)


MATCH_CASES = [  # (kind, seed, n0, n1, num_node)
    ("unit", 61, 3000, 3000, "all"), ("ties", 62, 1500, 1800, "all"), ("raw", 63, 2000, 2500, "all"),
    ("unit", 64, 1500, 1200, 2000), ("raw", 65, 700, 900, 1000),
]


def g6():
    """Key points are index-coded (x = row number), so the returned matched key points reveal the sampled rows and the
    arg-min indices; for an integer ``num_node`` the global ``np.random`` is seeded with the case's seed."""
    out = {"cases": np.array(json.dumps(MATCH_CASES))}
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for i, (kind, seed, n0, n1, num_node) in enumerate(MATCH_CASES):
            F0, F1 = gi.match_pair_case(kind, seed, n0, n1)
            k0 = np.zeros((1, n0, 3), np.float32); k0[0, :, 0] = np.arange(n0)
            k1 = np.zeros((1, n1, 3), np.float32); k1[0, :, 0] = np.arange(n1)
            m = Matcher(**{**KITTI_CFG, "num_node": num_node})
            np.random.seed(seed)
            sc, tc = m.match_pair(torch.from_numpy(k0), torch.from_numpy(k1), torch.from_numpy(F0)[None], torch.from_numpy(F1)[None])
            out[f"src{i}"] = sc[0, :, 0].numpy().astype(np.int32)
            out[f"tgt{i}"] = tc[0, :, 0].numpy().astype(np.int32)
    finally:
        torch.Tensor.cuda = real_cuda
    np.savez_compressed(os.path.join(HERE, "g6_match.npz"), **out)


LOSS_CASES = [  # (seed, n0, n1, n_pairs, num_pos, num_hn_samples)
    (71, 1500, 1400, 1300, 1000, 512),      # more positives than num_pos: the third draw happens
    (72, 900, 1100, 600, 5192, 2048),       # fewer: every positive is used; clouds smaller than num_hn_samples, so every anchor's
                                            # nearest candidate is its own partner, all negatives are masked and the reference's
                                            # mean over nothing is NaN (kept: that is what the trainer would log)
    (73, 300, 280, 2000, 256, 64),          # many duplicates among the positives, tiny candidate sets
]


def g7():
    from types import SimpleNamespace
    from lib.trainer import CorrespondenceExtensionTrainer, HardestContrastiveLossTrainer
    out = {"cases": np.array(json.dumps(LOSS_CASES))}
    me = SimpleNamespace(pos_thresh=0.1, neg_thresh=1.4)          # config.py:113-114 defaults
    torch.set_num_threads(1)                                      # index_add over duplicated rows: one thread = one summation order = a reproducible fixture
    for i, (seed, n0, n1, npairs, num_pos, nhn) in enumerate(LOSS_CASES):
        F0n, F1n, pairs = gi.loss_case(seed, n0, n1, npairs)
        res = []
        for cls in (CorrespondenceExtensionTrainer, HardestContrastiveLossTrainer):   # :935 and its twin :428
            F0 = torch.from_numpy(F0n).requires_grad_(True)
            F1 = torch.from_numpy(F1n).requires_grad_(True)
            np.random.seed(seed)
            pos, neg = cls.contrastive_hardest_negative_loss(me, F0, F1, torch.from_numpy(pairs), num_pos=num_pos, num_hn_samples=nhn)
            (pos + neg).backward()
            res.append((float(pos), float(neg), F0.grad.numpy().copy(), F1.grad.numpy().copy()))
        # the twins agree; the gradients only to rounding (index_add over duplicated rows sums in thread order)
        assert np.array_equal(res[0][:2], res[1][:2], equal_nan=True), (res[0][:2], res[1][:2])
        for a, b in ((res[0][2], res[1][2]), (res[0][3], res[1][3])):
            assert np.abs(a - b).max() <= 1e-7 * max(np.abs(a).max(), 1e-30), np.abs(a - b).max()
        out[f"pos{i}"], out[f"neg{i}"] = np.array(res[0][0], np.float64), np.array(res[0][1], np.float64)
        out[f"gF0_{i}"], out[f"gF1_{i}"] = res[0][2], res[0][3]
    torch.set_num_threads(8)
    np.savez_compressed(os.path.join(HERE, "g7_loss.npz"), **out)


LABEL_CASES = [(81, 3000, 3000, 1000), (82, 500, 800, 5000), (83, 1200, 37, 64)]   # (seed, n0, n1, num_corres)


def g8():
    """Inputs: the two smallest squared distances and the nearest index of every row of F0 among F1 (what pytorch3d's
    ``knn_points(K = 2)`` hands the trainer at :1060; computed by the test-side numpy restatement and STORED, inputs are data).
    Outputs: the reference's weights (:1066-1070 feed ``calculate_ratio_test`` the cosines ``1 - 0.5 d``) and its top-k."""
    from lib.trainer import CorrespondenceExtensionTrainer as Tr
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle.labels import knn2
    out = {"cases": np.array(json.dumps(LABEL_CASES))}
    for i, (seed, n0, n1, k) in enumerate(LABEL_CASES):
        F0, F1 = gi.nn_case(seed, n0, n1)
        idx, d1, d2 = knn2(F0, F1)
        dists = torch.from_numpy(np.stack([d1, d2], 1))[None]                # (1, P, 2) squared L2
        w = Tr.calculate_ratio_test(None, 1 - 0.5 * dists)                   # (1, P, 1)
        src, tgt, top = Tr.get_topk_matches(None, w, torch.from_numpy(idx)[None, :, None], k)
        out[f"d1_{i}"], out[f"d2_{i}"], out[f"idx{i}"] = d1, d2, idx.astype(np.int32)
        out[f"w{i}"] = w[0, :, 0].numpy()
        out[f"src{i}"], out[f"tgt{i}"], out[f"top{i}"] = src[0, :, 0].numpy().astype(np.int32), tgt[0, :, 0].numpy().astype(np.int32), top[0, :, 0].numpy()
    np.savez_compressed(os.path.join(HERE, "g8_labels.npz"), **out)


EVAL_CASES = [(91, 3000, 2600, 1000), (92, 400, 500, 1000), (93, 800, 800, -1)]    # (seed, n0, n1, subsample_size)


def g9():
    """Points are index-coded (x = row number) so the returned arrays reveal the drawn rows; the global ``np.random`` is
    seeded with the case's seed right before each call."""
    import scripts.test_kitti as tk
    out = {"cases": np.array(json.dumps(EVAL_CASES))}
    for i, (seed, n0, n1, sub) in enumerate(EVAL_CASES):
        F0, F1 = gi.nn_case(seed, n0, n1)
        x0 = np.zeros((n0, 3), np.float32); x0[:, 0] = np.arange(n0)
        x1 = np.zeros((n1, 3), np.float32); x1[:, 0] = np.arange(n1)
        np.random.seed(seed)
        a, b = tk.find_corr(torch.from_numpy(x0), torch.from_numpy(x1), torch.from_numpy(F0), torch.from_numpy(F1), subsample_size=sub)
        out[f"corr0_{i}"], out[f"corr1_{i}"] = a[:, 0].numpy().astype(np.int32), b[:, 0].numpy().astype(np.int32)
    # random_sample: n > N (permutation prefix), n < N (with replacement), n == N (identity); numpy and torch inputs
    for j, (n, N) in enumerate(((1000, 300), (200, 500), (64, 64))):
        pts = np.zeros((n, 3), np.float32); pts[:, 0] = np.arange(n)
        feats = np.arange(n * 4, dtype=np.float32).reshape(n, 4)
        np.random.seed(100 + j)
        p, f = tk.random_sample(pts, torch.from_numpy(feats), N)
        out[f"rs_p{j}"], out[f"rs_f{j}"] = np.asarray(p)[:, 0].astype(np.int32), f.numpy()
    T = gi.rigid(0.03, -0.02, 0.4, 7.0, -1.0, 0.3).astype(np.float32)
    pts = ((gi._u(95, 500, 3) - 0.5) * 60).astype(np.float32)
    tgt = (pts @ T[:3, :3].T + T[:3, 3] + 0.05 * gi._normal(96, 500, 3)).astype(np.float32)
    out["T"] = T
    out["applied"] = tk.apply_transform(torch.from_numpy(pts), torch.from_numpy(T)).numpy()
    # called like scripts/test_kitti.py:166 does - torch tensors in (apply_transform uses Tensor.t()), a list of floats out
    out["nn_dist"] = np.array(tk.evaluate_nn_dist(torch.from_numpy(pts), torch.from_numpy(tgt), torch.from_numpy(T)), np.float64)
    np.savez_compressed(os.path.join(HERE, "g9_eval.npz"), **out)


G10_CASES = [("Lowe", "Spherical", None), ("Lowe", "Similarity", [3, 17, 40]), ("Lowe", "None", None), ("None", "Spherical", None)]


def g10():
    """The reference's own ``match_and_filter_corr`` (lib/trainer.py:1025-1151) - re-collation with the collate biases, top-k in both
    directions, spherical and similarity filters - run unmodified on a batch of three pairs.  Two pytorch3d names it calls are absent
    here and are given bodies FOR THIS FIXTURE ONLY, neither of them the code under test:
      * ``pytorch3d.structures.Pointclouds``: a list -> zero-padded tensor holder (``features_padded``, ``num_points_per_cloud``);
      * ``knn_points(K = 2)``: returns STORED neighbours - the two smallest squared distances and the nearest index of every row,
        computed by the numpy restatement ``oracle.labels.knn2`` (the neighbour search itself is pinned elsewhere: G1 / G8 inputs) and
        padded with zeros like pytorch3d pads rows beyond a cloud's length.
    The similarity table is synthetic (``_inputs.dist_sim_table``; set on the trainer so that it does not read the reference's file)."""
    from types import SimpleNamespace
    import lib.trainer as lt
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle.labels import knn2

    class Pointclouds:
        def __init__(self, points, features=None):
            self.f = [torch.as_tensor(f) for f in features]

        def features_padded(self):
            n = max(len(f) for f in self.f)
            out = torch.zeros((len(self.f), n, self.f[0].shape[1]), dtype=self.f[0].dtype)
            for i, f in enumerate(self.f):
                out[i, :len(f)] = f
            return out

        def num_points_per_cloud(self):
            return torch.tensor([len(f) for f in self.f])

    C0s, F0s, C1s, F1s = gi.label_batch_case(101)
    stored = []
    for A, B in ((F0s, F1s), (F1s, F0s)):
        n = max(len(a) for a in A)
        d = np.zeros((len(A), n, 2), np.float32)
        ix = np.zeros((len(A), n, 2), np.int64)
        for i, (a, b) in enumerate(zip(A, B)):
            idx, d1, d2 = knn2(a, b)
            d[i, :len(a), 0], d[i, :len(a), 1], ix[i, :len(a), 0] = d1, d2, idx
        stored.append((torch.from_numpy(d), torch.from_numpy(ix)))
    out = {"cases": np.array(json.dumps(G10_CASES))}
    table = gi.dist_sim_table()
    real_pc, real_knn = lt.pytorch3d.structures.Pointclouds, getattr(lt, "knn_points", None)
    try:
        lt.pytorch3d.structures.Pointclouds = Pointclouds
        for i, (ff, sf, fd) in enumerate(G10_CASES):
            calls = []

            def knn_points(p1, p2, n1, n2, K=1):
                calls.append(K)
                d, ix = stored[len(calls) - 1]
                return d[:, :, :K].clone(), ix[:, :, :K].clone(), None

            lt.knn_points = knn_points
            me = SimpleNamespace(config=SimpleNamespace(similarity_thresh=0.3, pretraining_dataset="kitti"),
                                 dist_sim_map={k: torch.tensor(v) for k, v in table.items()})
            me.calculate_ratio_test = lambda d: lt.CorrespondenceExtensionTrainer.calculate_ratio_test(me, d)
            me.get_topk_matches = lambda d, ix, k: lt.CorrespondenceExtensionTrainer.get_topk_matches(me, d, ix, k)
            matches, unc = lt.CorrespondenceExtensionTrainer.match_and_filter_corr(
                me, [torch.from_numpy(c) for c in C0s], [torch.from_numpy(f) for f in F0s], [torch.from_numpy(c) for c in C1s],
                [torch.from_numpy(f) for f in F1s], radius=20, feature_filter=ff, spatial_filter=sf, frame_distance=fd)
            assert len(calls) == 2
            out[f"matches{i}"] = matches.numpy().astype(np.int64)
            for p, u in enumerate(unc):
                out[f"unc{i}_{p}"] = u.numpy().astype(np.int64)
    finally:
        lt.pytorch3d.structures.Pointclouds = real_pc
        if real_knn is not None:
            lt.knn_points = real_knn
    np.savez_compressed(os.path.join(HERE, "g10_match_filter.npz"), **out)


if __name__ == "__main__":
    for fn in (g1, g2, g3, g4, g5, g6, g7, g8, g9, g10):
        fn()
        print("wrote", fn.__name__)
