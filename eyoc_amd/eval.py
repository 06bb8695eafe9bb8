"""Feature-space nearest neighbours with the reference's call surface.

``find_nn_gpu`` / ``pdist`` mirror ``lib/eval.py:18-48`` and ``lib/metrics.py:22-29``; ``find_corr`` and
``random_sample`` mirror ``scripts/test_kitti.py:28-42,54-73`` (twin of ``find_corr`` at
``lib/trainer.py:405-419``).  ``find_correspondences`` is the alias named by the project brief.

The search runs in ``libeyoc_hip.so`` (``eyoc_knn1``): the reference materialises a
``[500, 5000, 32]`` difference tensor per chunk and synchronises after each; here nothing is
materialised and ``nn_max_n`` (a memory knob that never changes the result) is accepted and ignored.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

# "GemmL2" is not a reference name: it selects Matcher.match_pair's sqrt(2 - 2 <a,b> + 1e-6) (SC2_PCR.py:296-298)
_DIST = {"SquareL2": 0, "L2": 1, "GemmL2": 2}


def _cuda_f32(t, device=None) -> torch.Tensor:
    """fp32, contiguous, on the GPU (host inputs are uploaded - there is no CPU compute path)."""
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(np.asarray(t))
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise _lib.EyocError("no GPU visible: the EYOC hot path runs on MI355X only (no CPU fallback)")
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    return t.to(torch.float32).contiguous()


MAX_SEGMENTS = 128


def _segment_chunks(seg_a, seg_b):
    """The library takes at most ``MAX_SEGMENTS`` segments per launch: yield ``(a0, sa, sb, ns)`` per launch, with
    ``sa`` re-based to the chunk's first A row ``a0`` (outputs are per A row) and ``sb`` absolute."""
    nseg = len(seg_a) - 1
    for s0 in range(0, nseg, MAX_SEGMENTS):
        ns = min(MAX_SEGMENTS, nseg - s0)
        a0 = int(seg_a[s0])
        if int(seg_a[s0 + ns]) == a0:
            continue
        sa = (C.c_int32 * (ns + 1))(*[int(v) - a0 for v in seg_a[s0:s0 + ns + 1]])
        sb = (C.c_int32 * (ns + 1))(*[int(v) for v in seg_b[s0:s0 + ns + 1]])
        yield a0, sa, sb, ns


def knn1_segmented(A: torch.Tensor, B: torch.Tensor, seg_a, seg_b, dist_type="SquareL2", return_distance=True):
    """Independent 1-NN problems packed in one launch: rows ``seg_a[s]:seg_a[s+1]`` of ``A`` against rows
    ``seg_b[s]:seg_b[s+1]`` of ``B``.  Returns device tensors ``idx int64 [len(A)]`` (local to the B
    segment) and ``dist f32 [len(A)]``."""
    if dist_type not in _DIST:
        raise NotImplementedError('Not implemented')
    A, B = _cuda_f32(A), _cuda_f32(B, A.device if A.is_cuda else None)
    idx = torch.empty(A.shape[0], dtype=torch.int64, device=A.device)
    # no distances asked for: the library is told so (NULL) - it may then decide most rows by an MFMA score (knn.hip)
    dist = torch.empty(A.shape[0] if return_distance else 0, dtype=torch.float32, device=A.device)
    if A.shape[0] == 0:
        return (idx, dist) if return_distance else idx
    if A.shape[1] != B.shape[1]:
        raise ValueError("feature dimensions differ")
    with torch.cuda.device(A.device):
        for a0, sa, sb, ns in _segment_chunks(seg_a, seg_b):
            _lib.check(_lib.load().eyoc_knn1(_lib.ctx(A.device.index), _lib.ptr(A[a0:]), _lib.ptr(B), A.shape[1], sa, sb, ns,
                                             _DIST[dist_type], _lib.ptr(idx[a0:]), _lib.ptr(dist[a0:]) if return_distance else None,
                                             _lib.stream_ptr()),
                       "eyoc_knn1")
    return (idx, dist) if return_distance else idx


def gather_rows(F: torch.Tensor, sel: torch.Tensor, G: torch.Tensor | None = None, beta: float = 0.0) -> torch.Tensor:
    """``F[sel]`` in one launch (``eyoc_gather_rows``); with ``G`` ``[len(sel), C]`` the gathered rows are blended
    and re-normalised, ``(F[sel] + beta * G) / |.|`` - the descriptor mode of the synthetic benchmark."""
    F = _cuda_f32(F)
    sel = sel.to(F.device, torch.int64).contiguous()
    c = F.shape[1]
    out = torch.empty((sel.shape[0], c), dtype=torch.float32, device=F.device)
    if G is not None:
        G = _cuda_f32(G, F.device)
        if tuple(G.shape) != (sel.shape[0], c):
            raise ValueError("G must be [len(sel), C]")
    with torch.cuda.device(F.device):
        _lib.check(_lib.load().eyoc_gather_rows(_lib.ctx(F.device.index), _lib.ptr(F), F.stride(0), c, _lib.ptr(sel),
                                                sel.shape[0], _lib.ptr(G), C.c_float(beta), _lib.ptr(out),
                                                _lib.stream_ptr()), "eyoc_gather_rows")
    return out


def dotmax_segmented(A, B, seg_a, seg_b):
    """``(A @ B.T).max(dim=1)`` per segment without the matrix: ``(weight f32 [len(A)], idx int64 [len(A)])`` on the
    device (util/transform_estimation.py:131-133)."""
    A, B = _cuda_f32(A), _cuda_f32(B, A.device if A.is_cuda else None)
    if A.shape[1] != B.shape[1]:
        raise ValueError("feature dimensions differ")
    idx = torch.zeros(A.shape[0], dtype=torch.int64, device=A.device)
    w = torch.full((A.shape[0],), float("nan"), dtype=torch.float32, device=A.device)
    if A.shape[0]:
        with torch.cuda.device(A.device):
            for a0, sa, sb, ns in _segment_chunks(seg_a, seg_b):
                _lib.check(_lib.load().eyoc_dotmax(_lib.ctx(A.device.index), _lib.ptr(A[a0:]), _lib.ptr(B), A.shape[1], sa, sb,
                                                   ns, _lib.ptr(idx[a0:]), _lib.ptr(w[a0:]), _lib.stream_ptr()), "eyoc_dotmax")
    return w, idx


def find_nn_gpu(F0, F1, nn_max_n=-1, return_distance=False, dist_type='SquareL2'):
    """lib/eval.py:18-48.  Returns CPU tensors like the reference: ``inds int64 [N0]`` and, on request,
    ``dists f32 [N0,1]``."""
    F0 = _cuda_f32(F0)
    F1 = _cuda_f32(F1, F0.device)
    if F0.shape[1] != F1.shape[1]:
        raise ValueError("feature dimensions differ")
    if dist_type not in ("SquareL2", "L2"):          # lib/metrics.py:29
        raise NotImplementedError('Not implemented')
    idx, dist = knn1_segmented(F0, F1, [0, F0.shape[0]], [0, F1.shape[0]], dist_type)
    inds = idx.cpu()
    if return_distance:
        return inds, dist.unsqueeze(1).cpu()
    return inds


def pdist(A, B, dist_type='L2'):
    """lib/metrics.py:22-29 - dense ``[n,m]`` distance matrix on the device of ``A``."""
    if dist_type not in ("SquareL2", "L2"):
        raise NotImplementedError('Not implemented')
    A = _cuda_f32(A)
    B = _cuda_f32(B, A.device)
    out = torch.empty((A.shape[0], B.shape[0]), dtype=torch.float32, device=A.device)
    with torch.cuda.device(A.device):
        _lib.check(_lib.load().eyoc_pdist(_lib.ctx(A.device.index), _lib.ptr(A), A.shape[0], _lib.ptr(B), B.shape[0],
                                          A.shape[1], _DIST[dist_type], _lib.ptr(out), _lib.stream_ptr()), "eyoc_pdist")
    return out


def find_corr(xyz0, xyz1, F0, F1, subsample_size=-1, rng=None):
    """scripts/test_kitti.py:28-42.  ``rng`` (a ``numpy.random.Generator`` or ``RandomState``) replaces the
    reference's use of the global ``np.random`` so callers can make the draw reproducible."""
    rng = np.random if rng is None else rng
    subsample = len(F0) > subsample_size
    if subsample_size > 0 and subsample:
        N0 = min(len(F0), subsample_size)
        N1 = min(len(F1), subsample_size)
        inds0 = rng.choice(len(F0), N0, replace=False)
        inds1 = rng.choice(len(F1), N1, replace=False)
        F0, F1 = F0[inds0], F1[inds1]
    nn_inds = find_nn_gpu(F0, F1, nn_max_n=500)
    if subsample_size > 0 and subsample:
        return xyz0[inds0], xyz1[inds1[nn_inds]]
    return xyz0, xyz1[nn_inds]


find_correspondences = find_corr


def random_sample(pcd, feats, N, rng=None):
    """scripts/test_kitti.py:54-73 - exactly ``N`` points (permutation if n > N, with replacement if n < N)."""
    rng = np.random if rng is None else rng
    n1 = pcd.size(0) if isinstance(pcd, torch.Tensor) else pcd.shape[0]
    if n1 == N:
        return pcd, feats
    choice = rng.permutation(n1)[:N] if n1 > N else rng.choice(n1, N)
    return pcd[choice], feats[choice]
