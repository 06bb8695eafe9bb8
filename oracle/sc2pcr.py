"""CPU oracle of SC2-PCR registration as EYOC uses it (torch on CPU, fp32).

TEST INFRASTRUCTURE.  Pinned against golden vectors produced by the reference's own
``scripts.SC2_PCR.SC2_PCR.Matcher`` (``tests/golden/make_golden.py`` -> ``g4_sc2pcr.npz``).

Restates ``Matcher`` (scripts/SC2_PCR/SC2_PCR.py:7-413) under the KITTI constants of
scripts/SC2_PCR/config_json/config_KITTI.json:1-15.  Stage by stage:

  ``SC2_PCR`` (:307-384)      truncate to max_points; pairwise length-difference matrix
                              ``|d_src(i,j) - d_tgt(i,j)|``; soft first-order compatibility
                              ``clamp(1 - cross^2 / d_thre^2, 0)`` and two hard masks (< d_thre, < d_thre/2);
  ``leading_eigvec`` (:170-196)  <= num_iterations power steps from the all-ones vector, each
                              normalised by ``|v| + 1e-6``, early exit on ``allclose``;
  ``pick_seeds`` (:33-59)     non-maximum suppression of the eigenvector within nms_radius (source
                              space), top ``int(ratio * N)`` by score;
  second-order measure (:352-363)  ``(tight[seeds] @ tight) * hard[seeds]``;
  ``cal_seed_trans`` (:61-168)   per seed: top-k1 by SC2, local SC2 (first row of the local hard matrix
                              times the matrix), top-k2, local soft matrix with zero diagonal, power
                              iteration weights, weighted Kabsch, inlier count over all
                              correspondences, argmax;
  ``post_refinement`` (:238-278) <= 20 rounds: inliers under 1.2 m (KITTI branch), Cauchy weights
                              ``1 / (1 + (d / thr)^2)``, Kabsch; stop when the inlier count repeats.

``argsort`` ties (integer-valued SC2 counts, zero-padded scores) are implementation-defined in
the reference (unstable sort); this restatement breaks them towards the lower index.  Only the
final pose is compared with the golden vectors.
"""
from __future__ import annotations

import numpy as np
import torch

from .pose import rigid_transform_3d, transform

KITTI_CFG = dict(inlier_threshold=0.6, num_node=8000, use_mutual=False, d_thre=0.1,
                 num_iterations=20, ratio=0.2, nms_radius=0.6, max_points=8000, k1=30, k2=20)


def _desc_order(x, dim):
    return torch.argsort(x, dim=dim, descending=True, stable=True)


def pairwise_len(p):
    return torch.norm(p[:, :, None, :] - p[:, None, :, :], dim=-1)


class Matcher:
    def __init__(self, inlier_threshold=0.10, num_node="all", use_mutual=True, d_thre=0.1,
                 num_iterations=10, ratio=0.2, nms_radius=0.1, max_points=8000, k1=30, k2=20,
                 heatmap=False):
        self.inlier_threshold = inlier_threshold
        self.num_node = num_node
        self.use_mutual = use_mutual
        self.d_thre = d_thre
        self.num_iterations = num_iterations
        self.ratio = ratio
        self.max_points = max_points
        self.nms_radius = nms_radius
        self.k1 = k1
        self.k2 = k2

    # ------------------------------------------------------------------ :170-196
    def cal_leading_eigenvector(self, M, method="power"):
        v = torch.ones_like(M[:, :, 0:1])
        last = v
        for _ in range(self.num_iterations):
            v = torch.bmm(M, v)
            v = v / (torch.norm(v, dim=1, keepdim=True) + 1e-6)
            if torch.allclose(v, last):
                break
            last = v
        return v.squeeze(-1)

    # ------------------------------------------------------------------ :33-59
    def pick_seeds(self, dists, scores, R, max_num):
        assert scores.shape[0] == 1
        s = scores[0]
        # i survives iff no j within R has a strictly larger score
        dominated = (s[None, :] > s[:, None]) & (dists[0] < R)
        keep = (~dominated.any(dim=1)).float()
        return _desc_order(scores * keep[None, :], 1)[:, :max_num]

    # ------------------------------------------------------------------ :61-168
    def cal_seed_trans(self, seeds, SC2_measure, src, tgt):
        bs, n_seed, n_corr = SC2_measure.shape
        k1, k2 = self.k1, self.k2
        if k1 > n_corr:
            k1 = k2 = 4
        nn1 = _desc_order(SC2_measure, 2)[:, :, :k1]                       # [bs, S, k1]
        take = lambda pts, idx: pts[0][idx[0]][None]                          # bs == 1 gather
        s1, t1 = take(src, nn1), take(tgt, nn1)                               # [1, S, k1, 3]
        loc = lambda p: ((p[:, :, :, None, :] - p[:, :, None, :, :]) ** 2).sum(-1) ** 0.5
        cross = torch.abs(loc(s1) - loc(t1))
        hard = (cross < self.d_thre).float()
        local_sc2 = torch.matmul(hard[:, :, :1, :], hard)                     # [1, S, 1, k1]
        nn2 = _desc_order(local_sc2, 3)[:, :, 0, :k2]                         # [1, S, k2]
        gather2 = lambda p: torch.gather(p, 2, nn2[..., None].expand(-1, -1, -1, 3))
        s2, t2 = gather2(s1), gather2(t1)                                     # [1, S, k2, 3]
        cross = torch.abs(loc(s2) - loc(t2))
        soft = torch.clamp(1 - cross ** 2 / self.d_thre ** 2, min=0).reshape(-1, k2, k2).clone()
        ar = torch.arange(k2)
        soft[:, ar, ar] = 0
        w = self.cal_leading_eigenvector(soft).reshape(bs, -1, k2)
        w = (w / (w.sum(-1, keepdim=True) + 1e-6)).reshape(-1, k2)
        T = rigid_transform_3d(s2.reshape(-1, k2, 3), t2.reshape(-1, k2, 3), w).reshape(bs, -1, 4, 4)
        pred = torch.einsum("bsnm,bmk->bsnk", T[:, :, :3, :3], src.permute(0, 2, 1)) + T[:, :, :3, 3:4]
        dist = torch.norm(pred.permute(0, 1, 3, 2) - tgt[:, None, :, :], dim=-1)
        fitness = (dist < self.inlier_threshold).float().sum(-1)              # [bs, S]
        best = fitness.argmax(dim=1)
        return T[torch.arange(bs), best], fitness

    # ------------------------------------------------------------------ :238-278
    def post_refinement(self, T, src, tgt, it_num, weights=None):
        assert T.shape[0] == 1
        thr = 0.10 if self.inlier_threshold == 0.10 else 1.2
        prev = 0
        for _ in range(it_num):
            d = torch.norm(transform(src, T) - tgt, dim=-1)
            inl = (d < thr)[0]
            n_inl = int(inl.sum())
            if abs(n_inl - prev) < 1:
                break
            prev = n_inl
            T = rigid_transform_3d(src[:, inl, :], tgt[:, inl, :], 1 / (1 + (d / thr) ** 2)[:, inl])
        return T

    # ------------------------------------------------------------------ :280-305
    def match_pair(self, src_keypts, tgt_keypts, src_features, tgt_features, rng=None):
        n_src, n_tgt = src_features.shape[1], tgt_features.shape[1]
        if self.num_node == "all":
            si, ti = np.arange(n_src), np.arange(n_tgt)
        else:
            rng = rng or np.random.default_rng(0)
            si, ti = rng.choice(n_src, self.num_node), rng.choice(n_tgt, self.num_node)
        sd, td = src_features[:, si, :], tgt_features[:, ti, :]
        sk, tk = src_keypts[:, si, :], tgt_keypts[:, ti, :]
        dist = torch.sqrt(2 - 2 * (sd[0] @ td[0].T) + 1e-6)
        j = torch.argmin(dist, dim=1)
        return sk, tk[:, j]

    # ------------------------------------------------------------------ :307-384
    def SC2_PCR(self, src_keypts, tgt_keypts):
        src, tgt = src_keypts.float(), tgt_keypts.float()
        n = tgt.shape[1]
        if n > self.max_points:
            src, tgt, n = src[:, :self.max_points], tgt[:, :self.max_points], self.max_points
        src_len = pairwise_len(src)
        cross = torch.abs(src_len - pairwise_len(tgt))
        soft = torch.clamp(1.0 - cross ** 2 / self.d_thre ** 2, min=0)
        hard = (cross < self.d_thre).float()
        conf = self.cal_leading_eigenvector(soft)
        seeds = self.pick_seeds(src_len, conf, R=self.nms_radius, max_num=int(n * self.ratio))
        tight = (cross < self.d_thre / 2).float()
        sc2 = torch.matmul(tight[0][seeds[0]][None], tight) * hard[0][seeds[0]][None]
        T, fitness = self.cal_seed_trans(seeds, sc2, src, tgt)
        T = self.post_refinement(T, src, tgt, 20)
        return T, fitness

    # ------------------------------------------------------------------ :386-413
    def estimator(self, src_keypts, tgt_keypts, src_features, tgt_features, rng=None):
        sc, tc = self.match_pair(src_keypts, tgt_keypts, src_features, tgt_features, rng)
        T, fitness = self.SC2_PCR(sc, tc)
        d = torch.sum((transform(sc, T) - tc) ** 2, dim=-1) ** 0.5
        return T, (d < self.inlier_threshold).float(), sc, tc, fitness
