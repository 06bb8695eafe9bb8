"""CPU oracle of the pose solvers and the evaluation metrics (torch on CPU, fp32).

TEST INFRASTRUCTURE.  Pinned against golden vectors produced by the reference's own functions
(``tests/golden/make_golden.py`` -> ``g2_irls.npz``, ``g3_kabsch.npz``, ``g5_se3.npz``); the metric
formulas (G6) cannot be imported (``scripts/test_kitti.py`` needs open3d/ME at module top) and are
pinned by hand-computed cases.

Restated functions:
  * ``est_quad_linear_robust`` + helpers   util/transform_estimation.py:5-116
  * ``rigid_transform_3d``                 scripts/SC2_PCR/common.py:7-45
  * ``transform`` / ``integrate_trans``    scripts/SC2_PCR/utils/SE3.py:44-96
  * RTE / RRE / success                    scripts/test_kitti.py:187-211
"""
from __future__ import annotations

import numpy as np
import torch


# ----------------------------------------------------------------------------- IRLS small-angle solver
def euler_zyx(x):
    """``R = Rz(x[2]) Ry(x[1]) Rx(x[0])`` (util/transform_estimation.py:5-45)."""
    cx, sx = torch.cos(x[0]), torch.sin(x[0])
    cy, sy = torch.cos(x[1]), torch.sin(x[1])
    cz, sz = torch.cos(x[2]), torch.sin(x[2])
    one, zero = torch.ones(()), torch.zeros(())
    Rx = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx]).reshape(3, 3)
    Ry = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy]).reshape(3, 3)
    Rz = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one]).reshape(3, 3)
    return Rz.mm(Ry).mm(Rx)


def _linear_system(p0, p1, w):
    """Rows of the linearised residual ``p1 - (p0 + r x p0 + t)`` (util/transform_estimation.py:56-77):
    x-block ``[0, z, -y, 1, 0, 0]``, y-block ``[-z, 0, x, 0, 1, 0]``, z-block ``[y, -x, 0, 0, 0, 1]``,
    blocks stacked, every row and right-hand side scaled by the point weight."""
    n = p0.shape[0]
    x, y, z = p0[:, 0], p0[:, 1], p0[:, 2]
    o, l = torch.zeros(n), torch.ones(n)
    A = torch.cat([torch.stack([o, z, -y, l, o, o], 1),
                   torch.stack([-z, o, x, o, l, o], 1),
                   torch.stack([y, -x, o, o, o, l], 1)], 0)
    b = torch.cat([p1[:, 0] - x, p1[:, 1] - y, p1[:, 2] - z], 0).unsqueeze(1)
    w3 = w.reshape(-1, 1).repeat(3, 1)
    return A * w3, b * w3


def est_quad_linear_robust(pts0, pts1, weight=None, iters=20):
    """util/transform_estimation.py:89-116: 20 IRLS steps, ``par`` halves at i = 5, 10, 15,
    weights ``par / (|p0 - p1| + par)``, normal equations solved through an explicit inverse."""
    pts0, pts1 = pts0.float(), pts1.float()
    cur = pts0
    T = torch.eye(4)
    par = 1.0
    w = torch.ones(pts0.shape[0], 1) if weight is None else weight.float()
    for i in range(iters):
        if i > 0 and i % 5 == 0:
            par /= 2.0
        A, b = _linear_system(cur, pts1, w)
        x = torch.inverse(A.t().mm(A)).mm(A.t()).mm(b)
        Ti = torch.eye(4)
        Ti[:3, :3] = euler_zyx(x[:3, 0])
        Ti[:3, 3] = x[3:, 0]
        cur = torch.t(Ti[:3, :3] @ torch.t(cur)) + Ti[:3, 3]
        w = par / (torch.norm(cur - pts1, dim=1).unsqueeze(1) + par)
        T = Ti.mm(T)
    return T


def pose_estimation(F0, F1, xyz0, xyz1, inds=None):
    """util/transform_estimation.py:131-136 given the two feature matrices (the forwards of :127-130 are
    ``oracle.resunet.resunet_forward``): ``corr = F0 F1^T`` (dense), ``weight, inds = corr.max(1)``,
    ``est_quad_linear_robust(xyz0, xyz1[inds], weight)``.  ``inds`` overrides the arg-max (tests: compare the
    solver on identical correspondences when a near-tie flips an index).  Returns ``(T [4,4], weight [n,1], inds)``."""
    F0, F1 = torch.as_tensor(F0).float(), torch.as_tensor(F1).float()
    corr = F0.mm(F1.t())
    weight, arg = corr.max(dim=1)
    if inds is not None:
        arg = torch.as_tensor(inds).long()
        weight = corr[torch.arange(len(F0)), arg]
    weight = weight.unsqueeze(1)
    T = est_quad_linear_robust(torch.as_tensor(xyz0).float(), torch.as_tensor(xyz1).float()[arg, :], weight)
    return T, weight, arg


# ----------------------------------------------------------------------------- weighted Kabsch
def integrate_trans(R, t):
    """scripts/SC2_PCR/utils/SE3.py:75-96 (batched form)."""
    T = torch.eye(4).repeat(R.shape[0], 1, 1)
    T[:, :3, :3] = R
    T[:, :3, 3] = t.reshape(-1, 3)
    return T


def transform(pts, T):
    """scripts/SC2_PCR/utils/SE3.py:44-59: ``R p + t`` for ``[bs,n,3]`` / ``[bs,4,4]`` or unbatched."""
    if pts.dim() == 3:
        return (T[:, :3, :3] @ pts.permute(0, 2, 1) + T[:, :3, 3:4]).permute(0, 2, 1)
    return (T[:3, :3] @ pts.T + T[:3, 3:4]).T


def rigid_transform_3d(A, B, weights=None, weight_threshold=0):
    """scripts/SC2_PCR/common.py:7-45.  Weighted centroids with 1e-6 in the denominators,
    ``H = Am^T diag(w) Bm``, ``R = V diag(1,1,det(V U^T)) U^T``, ``t = cB - R cA``.
    (The reference thresholds ``weights`` in place; callers here never rely on that side effect.)"""
    A, B = A.float(), B.float()
    w = torch.ones_like(A[:, :, 0]) if weights is None else weights.float().clone()
    w[w < weight_threshold] = 0
    den = w.sum(1, keepdim=True)[:, :, None] + 1e-6
    cA = (A * w[:, :, None]).sum(1, keepdim=True) / den
    cB = (B * w[:, :, None]).sum(1, keepdim=True) / den
    Am, Bm = A - cA, B - cB
    H = Am.permute(0, 2, 1) @ (w[:, :, None] * Bm)
    U, S, V = torch.svd(H)
    d = torch.det(V @ U.permute(0, 2, 1))
    D = torch.eye(3).repeat(A.shape[0], 1, 1)
    D[:, 2, 2] = d
    R = V @ D @ U.permute(0, 2, 1)
    t = cB.permute(0, 2, 1) - R @ cA.permute(0, 2, 1)
    return integrate_trans(R, t)


# ----------------------------------------------------------------------------- metrics
def registration_errors(T_est, T_gt, rte_thresh=2.0, rre_thresh=5.0):
    """scripts/test_kitti.py:187-211 - RTE [m], RRE [rad] with the diagonal clamp, success flag."""
    T_est = np.asarray(T_est, np.float32)
    T_gt = np.asarray(T_gt, np.float32)
    rte = float(np.linalg.norm(T_est[:3, 3] - T_gt[:3, 3]))
    M = T_est[:3, :3].T @ T_gt[:3, :3]
    idx = np.arange(3)
    M[idx, idx] = np.minimum(np.float32(1.0), M[idx, idx])
    with np.errstate(invalid="ignore"):
        rre = float(np.arccos((np.trace(M) - 1) / 2))
    ok = bool(rte < rte_thresh and not np.isnan(rre) and rre < np.pi / 180 * rre_thresh)
    return rte, rre, ok


def apply_transform(pts, T):
    """scripts/test_kitti.py:44-47."""
    return pts @ T[:3, :3].T + T[:3, 3]


def evaluate_nn_dist(xyz0, xyz1, T_gt):
    """scripts/test_kitti.py:49-52."""
    x = apply_transform(np.asarray(xyz0), np.asarray(T_gt))
    return np.sqrt(((x - np.asarray(xyz1)) ** 2).sum(1) + 1e-6)
