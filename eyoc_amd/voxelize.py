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


def _optional_channels(name, arr, n_points, upper):
    """One optional per-point attribute of ``extract_features``: ``None`` or a ``[n_points, 3]`` array with no value
    above ``upper`` (the reference refuses colours / normals above 1, util/misc.py:47-57)."""
    if arr is None:
        return None
    a = np.asarray(arr)
    if a.ndim != 2 or a.shape != (n_points, 3):
        raise AssertionError(f"{name} must be [{n_points}, 3], got {tuple(a.shape)}")
    if bool(np.any(a > upper)):              # the reference's own test (np.any(rgb > 1)); any dtype, empty arrays included
        raise ValueError("Invalid color. Color must range from [0, 1]" if name == "rgb"
                         else "Invalid normal. Normal must range from [-1, 1]")
    return a


def extract_features(model, xyz, rgb=None, normal=None, voxel_size=0.05, device=None, skip_check=False, is_eval=True):
    """Voxelise one cloud, run ``model`` on it and return ``(xyz of the kept points, their features)`` - the one-call API
    of util/misc.py:21-93.  Input channels, in the reference's order: colour shifted to [-0.5, 0.5], normal halved, or a
    single column of ones when neither is given.  The de-duplication runs on the GPU first; only the kept rows of the
    attributes are shifted / scaled and uploaded."""
    if is_eval:
        model.eval()
    pts = np.asarray(xyz)
    if not skip_check:
        if pts.ndim != 2 or pts.shape[1] != 3:
            raise AssertionError(f"xyz must be [N, 3], got {tuple(pts.shape)}")
        rgb = _optional_channels("rgb", rgb, len(pts), 1.0)
        normal = _optional_channels("normal", normal, len(pts), 1.0)
    dev = torch.device("cuda:0") if device is None else torch.device(device)
    coords, kept = sparse_quantize(torch.from_numpy(np.ascontiguousarray(pts, np.float32)).to(dev), voxel_size)
    kept_host = kept.cpu().numpy()
    # arithmetic in the attribute's own dtype, then ONE rounding to fp32 (what `torch.tensor(rgb - 0.5, float32)` does)
    columns = [torch.as_tensor(np.asarray(a)[kept_host] * scale + shift, dtype=torch.float32, device=dev)
               for a, scale, shift in ((rgb, 1.0, -0.5), (normal, 0.5, 0.0)) if a is not None]
    feats = torch.cat(columns, dim=1) if columns else torch.ones((len(kept_host), 1), dtype=torch.float32, device=dev)
    return pts[kept_host], model(SparseTensor(feats, coordinates=coords)).F
