"""Label generation of EYOC's training loop on the GPU: the reference's
``match_and_filter_corr`` (lib/trainer.py:1025-1151) and the non-mutual branch of ``corr_through_registration``
(lib/trainer.py:1195-1218), with the same argument meaning.  Nearest neighbours, ratio weights, top-k and the
filters run in ``libeyoc_hip.so`` (``eyoc_knn2``, ``eyoc_lowe_topk``, ``eyoc_pair_filter``)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .eval import _cuda_f32, _segment_chunks


def knn2_segmented(A, B, seg_a, seg_b):
    """``knn_points(A, B, K=2)`` for independent segments: nearest index (int64, local to the B segment) and the
    two smallest squared distances of every row of ``A``; feature width 4 / 16 / 32 / 64 / 128."""
    A = _cuda_f32(A)
    B = _cuda_f32(B, A.device)
    if A.shape[1] != B.shape[1]:
        raise ValueError("feature dimensions differ")
    idx = torch.zeros(A.shape[0], dtype=torch.int64, device=A.device)
    d1 = torch.full((A.shape[0],), float("inf"), dtype=torch.float32, device=A.device)
    d2 = torch.full((A.shape[0],), float("inf"), dtype=torch.float32, device=A.device)
    if A.shape[0] == 0:
        return idx, d1, d2
    with torch.cuda.device(A.device):
        for a0, sa, sb, ns in _segment_chunks(seg_a, seg_b):      # at most 128 segments per launch
            _lib.check(_lib.load().eyoc_knn2(_lib.ctx(A.device.index), _lib.ptr(A[a0:]), _lib.ptr(B), A.shape[1], sa, sb, ns,
                                             _lib.ptr(idx[a0:]), _lib.ptr(d1[a0:]), _lib.ptr(d2[a0:]), _lib.stream_ptr()),
                       "eyoc_knn2")
    return idx, d1, d2


def lowe_topk(d1, d2, k):
    """calculate_ratio_test + get_topk_matches (lib/trainer.py:993-1016) on the two nearest squared distances:
    ``(idx_source int64 [k], weight f32 [k])``, largest weight first."""
    d1, d2 = _cuda_f32(d1), _cuda_f32(d2)
    k = min(int(k), d1.shape[0])
    idx = torch.empty(k, dtype=torch.int64, device=d1.device)
    w = torch.empty(k, dtype=torch.float32, device=d1.device)
    if k:
        with torch.cuda.device(d1.device):
            _lib.check(_lib.load().eyoc_lowe_topk(_lib.ctx(d1.device.index), _lib.ptr(d1), _lib.ptr(d2), d1.shape[0], k,
                                                  _lib.ptr(idx), _lib.ptr(w), _lib.stream_ptr()), "eyoc_lowe_topk")
    return idx, w


def _pair_filter(mode, P0, P1, i0, i1, T, radius):
    P0 = _cuda_f32(P0)
    P1 = _cuda_f32(P1, P0.device)
    i0 = i0.to(P0.device, torch.int64).contiguous()
    i1 = i1.to(P0.device, torch.int64).contiguous()
    m = i0.shape[0]
    out = torch.empty((m, 2), dtype=torch.int64, device=P0.device)
    n_out = torch.zeros(1, dtype=torch.int32, device=P0.device)
    Td = None if T is None else _cuda_f32(torch.as_tensor(np.asarray(T, np.float32)).reshape(16), P0.device)
    with torch.cuda.device(P0.device):
        _lib.check(_lib.load().eyoc_pair_filter(_lib.ctx(P0.device.index), mode, _lib.ptr(P0), _lib.ptr(P1), _lib.ptr(i0),
                                                _lib.ptr(i1), m, _lib.ptr(Td), C.c_float(radius), _lib.ptr(out),
                                                _lib.ptr(n_out), _lib.stream_ptr()), "eyoc_pair_filter")
    return out[:int(n_out.item())]


def spherical_filter(C0, C1, idx0, idx1, radius):
    """lib/trainer.py:1107-1110: the pairs whose two endpoints are both farther than ``radius`` from their sensor."""
    return _pair_filter(0, C0, C1, idx0, idx1, None, float(radius))


FRAME_TO_YGRID = {0: 1, 1: 1.5, 2: 2, 3: 2.5, 4: 2.5, 5: 2.5}      # lib/trainer.py:1136


def load_dist_sim_map(path):
    """``config/dist_sim_plot/<dataset>_distSimPlot.npz`` as the reference reads it (lib/trainer.py:1128-1132):
    ``{frame_index 0..5: float64 [xlim, ylim]}``."""
    maps = np.load(path, allow_pickle=True)["res"].tolist()
    return {i: np.asarray(maps[i], np.float64) for i in range(6)}


def similarity_filter(C0, C1, idx0, idx1, dist_sim_map, frame_distance, similarity_thresh=0.4):
    """lib/trainer.py:1118-1149 for one pair: keep the index pairs whose (min centre distance, centre-distance gap)
    cell of the distance-similarity table exceeds ``similarity_thresh``.  ``dist_sim_map``: ``{0..5: [xlim, ylim]}``
    (``load_dist_sim_map``); ``frame_distance``: the pair's frame gap (the table slice is ``clamp(gap // 5, 0, 5)``)."""
    frame_index = min(max(0, int(frame_distance) // 5), 5)
    table = np.ascontiguousarray(np.asarray(dist_sim_map[frame_index], np.float64))
    xlim, ylim = table.shape
    P0 = _cuda_f32(C0)
    P1 = _cuda_f32(C1, P0.device)
    i0 = idx0.to(P0.device, torch.int64).contiguous()
    i1 = idx1.to(P0.device, torch.int64).contiguous()
    m = i0.shape[0]
    out = torch.empty((m, 2), dtype=torch.int64, device=P0.device)
    n_out = torch.zeros(1, dtype=torch.int32, device=P0.device)
    td = torch.from_numpy(table).to(P0.device)
    with torch.cuda.device(P0.device):
        _lib.check(_lib.load().eyoc_pair_filter_similarity(
            _lib.ctx(P0.device.index), _lib.ptr(P0), _lib.ptr(P1), _lib.ptr(i0), _lib.ptr(i1), m, _lib.ptr(td), xlim, ylim,
            C.c_float(5.0), C.c_float(FRAME_TO_YGRID[frame_index]), C.c_double(similarity_thresh), _lib.ptr(out), _lib.ptr(n_out),
            _lib.stream_ptr()), "eyoc_pair_filter_similarity")
    return out[:int(n_out.item())]


def match_and_filter_corr(C_batch_0, F_batch_0, C_batch_1, F_batch_1, radius=20, feature_filter="Lowe",
                          spatial_filter="Spherical", frame_distance=None, num_corres=5000, dist_sim_map=None,
                          similarity_thresh=0.4):
    """lib/trainer.py:1025-1151.  Lists of per-cloud ``[n_i,3]`` coordinates and ``[n_i,d]`` features ->
    ``(matches int64 [N,2] on the CPU, with the collate biases; list of per-pair [M_i,2] device tensors)``.
    ``spatial_filter="Similarity"`` needs ``frame_distance`` (one gap per pair) and ``dist_sim_map`` (the reference
    reads it from ``config/dist_sim_plot/<pretraining_dataset>_distSimPlot.npz``: ``load_dist_sim_map``) with
    ``similarity_thresh`` = ``config.similarity_thresh``."""
    if feature_filter not in ("None", "Lowe"):
        raise AssertionError(feature_filter)
    if spatial_filter not in ("Spherical", "None", "Similarity"):
        raise AssertionError(spatial_filter)
    if spatial_filter == "Similarity" and (dist_sim_map is None or frame_distance is None):
        raise ValueError('spatial_filter="Similarity" needs dist_sim_map and frame_distance')
    F0s = [_cuda_f32(f) for f in F_batch_0]
    dev = F0s[0].device
    F1s = [_cuda_f32(f, dev) for f in F_batch_1]
    n0 = [f.shape[0] for f in F0s]
    n1 = [f.shape[0] for f in F1s]
    seg0 = np.concatenate([[0], np.cumsum(n0)])
    seg1 = np.concatenate([[0], np.cumsum(n1)])
    A, B = torch.cat(F0s), torch.cat(F1s)
    i12, d1a, d2a = knn2_segmented(A, B, seg0, seg1)          # both directions, all pairs of the batch in one launch each
    i21, d1b, d2b = knn2_segmented(B, A, seg1, seg0)
    k1, k2 = min(num_corres, min(n0)), min(num_corres, min(n1))
    idx1, idx2 = [], []
    for p in range(len(F0s)):
        a0, a1, b0, b1 = int(seg0[p]), int(seg0[p + 1]), int(seg1[p]), int(seg1[p + 1])
        if feature_filter == "Lowe":
            s12, _ = lowe_topk(d1a[a0:a1], d2a[a0:a1], k1)
            s21, _ = lowe_topk(d1b[b0:b1], d2b[b0:b1], k2)
        else:                                                  # weights = the nearest distance itself (:1072-1073)
            s12 = torch.argsort(-d1a[a0:a1].double(), stable=True)[:k1]
            s21 = torch.argsort(-d1b[b0:b1].double(), stable=True)[:k2]
        t12, t21 = i12[a0:a1][s12], i21[b0:b1][s21]
        idx1.append(torch.cat([s12, t21]))
        idx2.append(torch.cat([t12, s21]))
    matches = torch.cat([torch.stack([a + int(seg0[p]), b + int(seg1[p])], 1) for p, (a, b) in enumerate(zip(idx1, idx2))])
    uncollated = []
    for p in range(len(F0s)):
        if spatial_filter == "None":
            uncollated.append(torch.stack([idx1[p], idx2[p]], 1))
        elif spatial_filter == "Similarity":
            uncollated.append(similarity_filter(C_batch_0[p], C_batch_1[p], idx1[p], idx2[p], dist_sim_map, frame_distance[p],
                                                similarity_thresh))
        else:
            uncollated.append(spherical_filter(C_batch_0[p], C_batch_1[p], idx1[p], idx2[p], radius))
    return matches.cpu(), uncollated


def apply_pose(T, P):
    """``R p + t`` with every fp32 operation rounded separately, left to right (the arithmetic ``eyoc_pair_filter``
    uses for its residual), so the nearest neighbour below and the filter see the same posed points."""
    T = torch.as_tensor(np.asarray(T, np.float32)).to(P.device)
    cols = [((T[r, 0] * P[:, 0] + T[r, 1] * P[:, 1]) + T[r, 2] * P[:, 2]) + T[r, 3] for r in range(3)]
    return torch.stack(cols, 1)


def correspondences_under_pose(pcd0, pcd1, T, n_sample=5000, max_dist=2.0, pos_sel=None, generator=None):
    """lib/trainer.py:1195-1218 for one pair: nearest ``pcd1`` point of every posed ``pcd0`` point, a random subset
    of at most ``n_sample`` of them (``torch.randperm`` like the reference; pass ``pos_sel`` to fix the draw), kept
    where the residual under ``T`` is below ``max_dist``.  Returns ``int64 [M,2]`` on the device."""
    P0 = _cuda_f32(pcd0)
    P1 = _cuda_f32(pcd1, P0.device)
    q = apply_pose(T, P0)
    pad = lambda P: torch.cat([P, torch.zeros((P.shape[0], 1), device=P.device)], 1).contiguous()
    idx, _, _ = knn2_segmented(pad(q), pad(P1), [0, P0.shape[0]], [0, P1.shape[0]])
    if pos_sel is None:
        pos_sel = torch.randperm(P0.shape[0], generator=generator)[:min(P0.shape[0], n_sample)]
    sel = torch.as_tensor(pos_sel).to(P0.device, torch.int64)
    return _pair_filter(1, P0, P1, sel, idx[sel], T, float(max_dist))
