"""Pose solvers with the reference's names.

``est_quad_linear_robust`` / ``pose_estimation`` mirror ``util/transform_estimation.py:89-144``;
``rigid_transform_3d`` mirrors ``scripts/SC2_PCR/common.py:7-45``; ``transform`` / ``integrate_trans``
mirror ``scripts/SC2_PCR/utils/SE3.py:44-96``.  ``estimate_transform`` is the alias named by the brief.
The reductions and the 3x3 SVD / 6x6 solve run in ``libeyoc_hip.so`` (fp64 accumulation on the GPU,
no host hop).
"""
from __future__ import annotations

import torch

from . import _lib
from .eval import _cuda_f32


def est_quad_linear_robust(pts0, pts1, weight=None, iters=20):
    """util/transform_estimation.py:89-116 -> ``T f32 [4,4]`` on the device of ``pts0``."""
    src_dev = pts0.device
    p0 = _cuda_f32(pts0)
    p1 = _cuda_f32(pts1, p0.device)
    if p0.shape != p1.shape or p0.dim() != 2 or p0.shape[1] != 3:
        raise ValueError("pts0 and pts1 must both be [N,3]")
    w = None if weight is None else _cuda_f32(weight, p0.device).reshape(-1)
    if w is not None and w.numel() != p0.shape[0]:
        raise ValueError("weight must have one entry per point")
    T = torch.empty((4, 4), dtype=torch.float32, device=p0.device)
    with torch.cuda.device(p0.device):
        _lib.check(_lib.load().eyoc_irls_quad(_lib.ctx(p0.device.index), _lib.ptr(p0), _lib.ptr(p1), _lib.ptr(w),
                                              p0.shape[0], int(iters), _lib.ptr(T), _lib.stream_ptr()), "eyoc_irls_quad")
    return T.to(src_dev)


estimate_transform = est_quad_linear_robust


def rigid_transform_3d(A, B, weights=None, weight_threshold=0):
    """scripts/SC2_PCR/common.py:7-45: ``A,B [bs,n,3]``, ``weights [bs,n]`` -> ``[bs,4,4]`` with ``B ~ R A + t``."""
    src_dev = A.device
    a = _cuda_f32(A)
    b = _cuda_f32(B, a.device)
    if a.dim() != 3 or a.shape != b.shape or a.shape[2] != 3:
        raise ValueError("A and B must both be [bs,n,3]")
    w = None
    if weights is not None:
        if weight_threshold > 0:
            weights[weights < weight_threshold] = 0      # in place, as the reference does (:20)
        w = _cuda_f32(weights, a.device)
    bs, n = a.shape[0], a.shape[1]
    T = torch.empty((bs, 4, 4), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().eyoc_kabsch_batched(_lib.ctx(a.device.index), _lib.ptr(a), _lib.ptr(b), _lib.ptr(w), bs,
                                                   n, _lib.ptr(T), _lib.stream_ptr()), "eyoc_kabsch_batched")
    return T.to(src_dev)


def transform(pts, trans):
    """scripts/SC2_PCR/utils/SE3.py:44-59."""
    if len(pts.shape) == 3:
        return (trans[:, :3, :3] @ pts.permute(0, 2, 1) + trans[:, :3, 3:4]).permute(0, 2, 1)
    return (trans[:3, :3] @ pts.T + trans[:3, 3:4]).T


def integrate_trans(R, t):
    """scripts/SC2_PCR/utils/SE3.py:75-96 (tensor forms)."""
    if len(R.shape) == 3:
        T = torch.eye(4, device=R.device)[None].repeat(R.shape[0], 1, 1)
        T[:, :3, :3] = R
        T[:, :3, 3:4] = t.view([-1, 3, 1])
    else:
        T = torch.eye(4, device=R.device)
        T[:3, :3] = R
        T[:3, 3:4] = t
    return T


def pose_estimation(model, device, xyz0, xyz1, coord0, coord1, feats0, feats1, return_corr=False):
    """util/transform_estimation.py:119-144: two forwards, inner-product arg-max as weight + index,
    IRLS.  The reference materialises the full ``[N0,N1]`` correlation matrix (3.6 GB at 30k voxels);
    it is only returned here when ``return_corr`` asks for it."""
    from .sparse_tensor import SparseTensor
    F0 = model(SparseTensor(feats0.to(device), coordinates=coord0.to(device))).F
    F1 = model(SparseTensor(feats1.to(device), coordinates=coord1.to(device))).F
    if return_corr:                              # the reference's dense form, only on request
        corr = F0.mm(F1.t())
        weight, inds = corr.max(dim=1)
    else:                                        # streaming arg-max of the same inner products (eyoc_dotmax)
        from .eval import dotmax_segmented
        weight, inds = dotmax_segmented(F0, F1, [0, F0.shape[0]], [0, F1.shape[0]])
    weight = weight.unsqueeze(1).cpu()
    xyz1_corr = xyz1[inds.cpu(), :]
    trans = est_quad_linear_robust(xyz0, xyz1_corr, weight)
    if return_corr:
        return trans, weight, corr
    return trans, weight
