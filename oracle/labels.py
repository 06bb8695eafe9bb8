"""CPU oracle of the label-generation glue (numpy).  TEST INFRASTRUCTURE - never imported by ``eyoc_amd``.

**Parity unpinned**: the functions restated here live in ``lib/trainer.py`` of the reference, which cannot be
imported in this image (MinkowskiEngine, pytorch3d and open3d are imported at module level and are absent), and
the reference has no test for them.  They are restated from the source, citing the lines they follow:

  * ``knn2``                     pytorch3d ``knn_points(..., K=2)`` as called at lib/trainer.py:1060-1061
                                 (squared L2, neighbours sorted by distance)
  * ``lowe_weights``             calculate_ratio_test, lib/trainer.py:993-1010, on the cosines of :1066-1070
  * ``topk_matches``             get_topk_matches, lib/trainer.py:1012-1016 (``torch.topk``: largest first)
  * ``match_and_filter_corr``    lib/trainer.py:1025-1151 (feature_filter "Lowe" / "None", spatial_filter
                                 "Spherical" / "None"; the "Similarity" filter needs the reference's
                                 ``config/dist_sim_plot/*.npz`` tables and is out of scope)
  * ``correspondences_under_pose``  the non-mutual branch of corr_through_registration, lib/trainer.py:1195-1218

Arithmetic shared bit-for-bit with the HIP kernels (``eyoc_knn2``, ``eyoc_lowe_topk``, ``eyoc_pair_filter``): fp32,
every operation rounded separately, in the order written here; distances as in ``oracle/matching.py``; ties of the
nearest neighbour go to the lowest index, ties of the top-k keep query order (``torch.topk`` leaves both open).
"""
from __future__ import annotations

import numpy as np

from .matching import sqdist_rows

F32 = np.float32


def knn2(A, B, chunk=256):
    """Nearest index (int64) and the two smallest squared distances of every row of A among the rows of B."""
    A = np.ascontiguousarray(A, F32)
    B = np.ascontiguousarray(B, F32)
    idx = np.zeros(len(A), np.int64)
    d1 = np.full(len(A), np.inf, F32)
    d2 = np.full(len(A), np.inf, F32)
    if len(B) == 0:
        return idx, d1, d2
    for s in range(0, len(A), chunk):
        D = sqdist_rows(A[s:s + chunk], B)
        j = np.argmin(D, axis=1)
        r = np.arange(len(j))
        idx[s:s + chunk] = j
        d1[s:s + chunk] = D[r, j]
        if B.shape[0] > 1:
            D[r, j] = np.inf
            d2[s:s + chunk] = D.min(axis=1)
    return idx, d1, d2


def lowe_weights(d1, d2):
    """:1066-1070 then :993-1010: cosine = 1 - 0.5 d; x = clamp(1 - cosine, 1e-9); weight = 1 - x0 / x1."""
    d1, d2 = np.asarray(d1, F32), np.asarray(d2, F32)
    c1 = F32(1) - F32(0.5) * d1
    c2 = F32(1) - F32(0.5) * d2
    x1 = np.maximum(F32(1) - c1, F32(1e-9))
    x2 = np.maximum(F32(1) - c2, F32(1e-9))
    with np.errstate(invalid="ignore", divide="ignore"):
        return (F32(1) - x1 / x2).astype(F32)


def topk_matches(weights, idx, k):
    """:1012-1016: the k largest weights (largest first, ties in query order) -> (idx_source, idx_target, weight)."""
    k = min(k, len(weights))
    order = np.argsort(-weights.astype(np.float64), kind="stable")[:k]
    return order.astype(np.int64), idx[order], weights[order]


def norm3(P):
    P = np.asarray(P, F32)
    return np.sqrt((P[:, 0] * P[:, 0] + P[:, 1] * P[:, 1]) + P[:, 2] * P[:, 2])


def apply_pose(T, P):
    """R p + t with every fp32 operation rounded separately, left to right (shared with the kernel)."""
    T = np.asarray(T, F32)
    P = np.asarray(P, F32)
    out = np.empty_like(P)
    for r in range(3):
        out[:, r] = ((T[r, 0] * P[:, 0] + T[r, 1] * P[:, 1]) + T[r, 2] * P[:, 2]) + T[r, 3]
    return out


FRAME_TO_YGRID = {0: 1, 1: 1.5, 2: 2, 3: 2.5, 4: 2.5, 5: 2.5}


def similarity_mask(C0, C1, a, b, dist_sim_map, frame_distance, similarity_thresh=0.4):
    """lib/trainer.py:1118-1149: the "Similarity" spatial filter of one pair.  ``d0, d1`` = centre distances of the
    two endpoints (fp32); table coordinates ``(min(d0, d1) / 5).long()`` and ``(|d0 - d1| / ygrid).long()`` clamped
    into the table, which is indexed ``[gap cell, min-distance cell]`` and compared in float64."""
    d0 = norm3(np.asarray(C0)[a])
    d1 = norm3(np.asarray(C1)[b])
    gap = np.abs(d0 - d1)
    dmin = np.minimum(d0, d1)
    frame_index = min(max(0, int(frame_distance) // 5), 5)
    table = np.asarray(dist_sim_map[frame_index], np.float64)
    xlim, ylim = table.shape
    c0 = (dmin / F32(5)).astype(np.int64)                               # fp32 division, truncation like .long()
    c1 = (gap / F32(FRAME_TO_YGRID[frame_index])).astype(np.int64)
    c0 = np.clip(c0, 0, ylim - 1)
    c1 = np.clip(c1, 0, xlim - 1)
    return table[c1, c0] > similarity_thresh


def match_and_filter_corr(C_batch_0, F_batch_0, C_batch_1, F_batch_1, radius=20, feature_filter="Lowe",
                          spatial_filter="Spherical", num_corres=5000, frame_distance=None, dist_sim_map=None,
                          similarity_thresh=0.4):
    """lib/trainer.py:1025-1151 -> (matches int64 [N,2] with the collate biases, list of per-pair [M_i,2])."""
    assert feature_filter in ("None", "Lowe") and spatial_filter in ("Spherical", "None", "Similarity")
    n1 = min(num_corres, min(len(f) for f in F_batch_0))
    n2 = min(num_corres, min(len(f) for f in F_batch_1))
    m1, m2 = [], []
    for F0, F1 in zip(F_batch_0, F_batch_1):
        i12, d1a, d2a = knn2(F0, F1)
        i21, d1b, d2b = knn2(F1, F0)
        w1 = lowe_weights(d1a, d2a) if feature_filter == "Lowe" else d1a
        w2 = lowe_weights(d1b, d2b) if feature_filter == "Lowe" else d1b
        s12, t12, _ = topk_matches(w1, i12, n1)          # cloud-0 query -> cloud-1 neighbour
        s21, t21, _ = topk_matches(w2, i21, n2)          # cloud-1 query -> cloud-0 neighbour
        m1.append(np.concatenate([s12, t21]))
        m2.append(np.concatenate([t12, s21]))
    bias1 = np.cumsum([0] + [len(f) for f in F_batch_0][:-1])
    bias2 = np.cumsum([0] + [len(f) for f in F_batch_1][:-1])
    matches = np.concatenate([np.stack([a + b1, b + b2], 1) for a, b, b1, b2 in zip(m1, m2, bias1, bias2)])
    uncollated = []
    for p, (C0, C1, a, b) in enumerate(zip(C_batch_0, C_batch_1, m1, m2)):
        if spatial_filter == "None":
            mask = np.ones(len(a), bool)
        elif spatial_filter == "Similarity":
            mask = similarity_mask(C0, C1, a, b, dist_sim_map, frame_distance[p], similarity_thresh)
        else:
            mask = (norm3(np.asarray(C0)[a]) > F32(radius)) & (norm3(np.asarray(C1)[b]) > F32(radius))
        uncollated.append(np.stack([a[mask], b[mask]], 1))
    return matches, uncollated


def correspondences_under_pose(pcd0, pcd1, T, pos_sel, max_dist=2.0):
    """lib/trainer.py:1195-1218 for one pair: nearest cloud-1 point of every posed cloud-0 point, the sampled
    subset ``pos_sel`` (the reference draws ``torch.randperm(n)[:5000]``), kept where the residual is < max_dist."""
    q = apply_pose(T, pcd0)
    pad = lambda P: np.concatenate([np.asarray(P, F32), np.zeros((len(P), 1), F32)], 1)
    idx, _, _ = knn2(pad(q), pad(pcd1))
    sel = np.asarray(pos_sel, np.int64)
    res = q[sel] - np.asarray(pcd1, F32)[idx[sel]]
    keep = norm3(res) < F32(max_dist)
    return np.stack([sel[keep], idx[sel][keep]], 1)
