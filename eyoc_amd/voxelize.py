"""Voxelisation and the one-call feature extractor with the reference's names.

``sparse_quantize`` mirrors the way the loaders call ``ME.utils.sparse_quantize(xyz / voxel_size,
return_index=True)`` (lib/data_loaders.py:940-943), ``voxelize`` adds the ``floor(...).int()`` +
ones-feature step of lib/data_loaders.py:969-972, and ``extract_features`` mirrors util/misc.py:21-93.
The hash-grid de-duplication runs in ``libeyoc_hip.so`` (``eyoc_voxelize``).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .sparse_tensor import SparseTensor


def sparse_quantize(xyz, voxel_size: float, batch_index: int = 0):
    """First point of every occupied ``voxel_size`` voxel, in input order.

    ``xyz``: ``[N,3]`` (or ``[N,4]`` KITTI xyzr) float32, numpy or torch.  Returns device tensors
    ``(coords int32 [M,4] = (batch, floor(p / voxel)), sel int64 [M])`` with ``sel`` ascending."""
    t = xyz if isinstance(xyz, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(xyz, np.float32))
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise _lib.EyocError("no GPU visible: the EYOC hot path runs on MI355X only (no CPU fallback)")
        t = t.cuda()
    t = t.to(torch.float32).contiguous()
    if t.dim() != 2 or t.shape[1] not in (3, 4):
        raise ValueError("xyz must be [N,3] or [N,4]")
    n = t.shape[0]
    lib = _lib.load()
    sel = torch.empty(max(n, 1), dtype=torch.int32, device=t.device)
    coords = torch.empty((max(n, 1), 4), dtype=torch.int32, device=t.device)
    if n == 0:
        return coords[:0], sel[:0].long()
    n_out = C.c_int(0)
    with torch.cuda.device(t.device):
        ws = _lib.workspace(lib.eyoc_voxelize_workspace_bytes(n), t.device)
        _lib.check(lib.eyoc_voxelize(_lib.ctx(t.device.index), _lib.ptr(t), n, t.shape[1], float(voxel_size), int(batch_index),
                                     _lib.ptr(sel), _lib.ptr(coords), C.byref(n_out), _lib.ptr(ws), ws.numel(),
                                     _lib.stream_ptr()), "eyoc_voxelize")
    m = n_out.value
    return coords[:m], sel[:m].long()


def voxelize(xyz, voxel_size: float, batch_index: int = 0):
    """``(xyz[sel], coords, feats = ones [M,1])`` - the per-cloud part of lib/data_loaders.py:936-979."""
    t = xyz if isinstance(xyz, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(xyz, np.float32))
    coords, sel = sparse_quantize(t, voxel_size, batch_index)
    pts = t.to(coords.device)[sel][:, :3]
    return pts, coords, torch.ones((len(sel), 1), dtype=torch.float32, device=coords.device)


def extract_features(model, xyz, rgb=None, normal=None, voxel_size=0.05, device=None, skip_check=False, is_eval=True):
    """util/misc.py:21-93 - voxelise one cloud, run the model, return ``(xyz[inds], features)``."""
    if is_eval:
        model.eval()
    xyz = np.asarray(xyz)
    if not skip_check:
        assert xyz.shape[1] == 3
        N = xyz.shape[0]
        if rgb is not None:
            assert N == len(rgb) and rgb.shape[1] == 3
            if np.any(rgb > 1):
                raise ValueError('Invalid color. Color must range from [0, 1]')
        if normal is not None:
            assert N == len(normal) and normal.shape[1] == 3
            if np.any(normal > 1):
                raise ValueError('Invalid normal. Normal must range from [-1, 1]')
    if device is None:
        device = torch.device("cuda:0")
    feats = []
    if rgb is not None:
        feats.append(rgb - 0.5)
    if normal is not None:
        feats.append(normal / 2)
    if rgb is None and normal is None:
        feats.append(np.ones((len(xyz), 1)))
    feats = np.hstack(feats)
    coords, inds = sparse_quantize(torch.from_numpy(xyz.astype(np.float32)).to(device), voxel_size)
    inds_h = inds.cpu().numpy()
    stensor = SparseTensor(torch.tensor(feats[inds_h], dtype=torch.float32, device=device), coordinates=coords)
    return xyz[inds_h], model(stensor).F
