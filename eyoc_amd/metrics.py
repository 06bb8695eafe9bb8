"""Evaluation metrics of the reference's test loop (scripts/test_kitti.py:44-52,187-211)."""
from __future__ import annotations

import numpy as np


def apply_transform(pts, trans):
    """scripts/test_kitti.py:44-47."""
    return pts @ trans[:3, :3].T + trans[:3, 3]


def evaluate_nn_dist(xyz0, xyz1, T_gth):
    """scripts/test_kitti.py:49-52."""
    xyz0 = apply_transform(np.asarray(xyz0), np.asarray(T_gth))
    return np.sqrt(((xyz0 - np.asarray(xyz1)) ** 2).sum(1) + 1e-6).tolist()


def registration_errors(T_est, T_gth, rte_thresh=2.0, rre_thresh=5.0):
    """RTE [m], RRE [rad] with the diagonal clamp that keeps ``arccos`` finite, and the success flag
    ``RTE < rte_thresh and RRE < rre_thresh deg`` (scripts/test_kitti.py:187-211)."""
    T_est = np.asarray(T_est, np.float32)
    T_gth = np.asarray(T_gth, np.float32)
    rte = float(np.linalg.norm(T_est[:3, 3] - T_gth[:3, 3]))
    M = T_est[:3, :3].T @ T_gth[:3, :3]
    d = np.arange(3)
    M[d, d] = np.minimum(np.float32(1.0), M[d, d])
    with np.errstate(invalid="ignore"):
        rre = float(np.arccos((np.trace(M) - 1) / 2))
    ok = bool(rte < rte_thresh and not np.isnan(rre) and rre < np.pi / 180 * rre_thresh)
    return rte, rre, ok


class AverageMeter:
    """lib/timer.py:5-25 (mean and variance of a running series)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.sq_sum = self.var = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
        self.sq_sum += val ** 2 * n
        self.var = self.sq_sum / self.count - self.avg ** 2
