"""Registration back-ends: feature-matching RANSAC and SC2-PCR, with the reference's call surface.

* ``registration_ransac_based_on_feature_matching`` stands in for the Open3D call at
  ``scripts/test_kitti.py:169-177`` (same positional meaning: source/target points, source/target
  features, mutual_filter, max_correspondence_distance, ransac_n = 4, edge-length + distance
  checkers, ``RANSACConvergenceCriteria(max_iteration, confidence)``).
* ``Matcher`` mirrors ``scripts/SC2_PCR/SC2_PCR.py:7-413`` (constructor, ``match_pair``, ``SC2_PCR``,
  ``estimator``).
Both run in ``libeyoc_hip.so``.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from .eval import _cuda_f32, knn1_segmented
from .transform_estimation import transform


@dataclass
class RegistrationResult:
    """The fields of ``open3d.pipelines.registration.RegistrationResult`` the harness reads."""
    transformation: np.ndarray      # float64 [4,4]
    fitness: float
    inlier_rmse: float
    inliers: int = 0
    best_hypothesis: int = -1
    survivors: int = 0


def ransac_from_correspondences(src, tgt, corr_tgt, max_correspondence_distance, max_iteration=4000000, seed=0,
                                edge_similarity=0.9, as_device_result=False):
    """``src [n,3]``, ``tgt [m,3]``, ``corr_tgt int64 [n]``: 4-point RANSAC over the correspondences
    ``(i, corr_tgt[i])``.  Returns a ``RegistrationResult`` (one device->host copy of 84 bytes)."""
    s = _cuda_f32(src)
    t = _cuda_f32(tgt, s.device)
    c = corr_tgt.to(s.device, torch.int64).contiguous()
    n = s.shape[0]
    res = ransac_batched_from_correspondences(s, t, c, [0, n], [0, 0], max_correspondence_distance, max_iteration, seed,
                                              edge_similarity)[0]
    if as_device_result:
        return res
    return decode_ransac_result(res, n)


def ransac_batched_from_correspondences(src, tgt, corr_tgt, seg_src, seg_tgt, max_correspondence_distance,
                                        max_iteration=4000000, seed=0, edge_similarity=0.9, workspace_budget=None):
    """All pairs of a batch in a few launches: ``src [N,3]`` / ``corr_tgt int64 [N]`` hold the pairs back to back
    (pair ``b`` = rows ``seg_src[b]:seg_src[b+1]``), ``tgt [M,3]`` likewise with ``seg_tgt``; ``corr_tgt`` indexes
    INSIDE the pair's target segment.  Pair ``b`` uses ``seed + b``.  Returns the ``[P, 84]`` byte tensor of
    ``eyoc_ransac_result`` records on the device (decode with ``decode_ransac_result``)."""
    s = _cuda_f32(src)
    t = _cuda_f32(tgt, s.device)
    c = corr_tgt.to(s.device, torch.int64).contiguous()
    P = len(seg_src) - 1
    ss = (C.c_int32 * (P + 1))(*[int(v) for v in seg_src])
    st = (C.c_int32 * (P + 1))(*[int(v) for v in seg_tgt])
    p = _lib.RansacParams(float(max_correspondence_distance), float(edge_similarity), int(max_iteration), int(seed))
    res = torch.empty((P, C.sizeof(_lib.RansacResult)), dtype=torch.uint8, device=s.device)
    lib = _lib.load()
    with torch.cuda.device(s.device):
        # the scratch (survivor lists: 144 MB per pair at 4 M hypotheses) is the caller's, i.e. torch's caching allocator's:
        # the largest launch chunk that fits ``workspace_budget`` bytes (default: a quarter of what the device and the
        # allocator's cache have free, at most 16 GB); any chunk size gives the same results
        budget = int(workspace_budget) if workspace_budget else _ransac_budget(s.device)
        ws = _lib.workspace(lib.eyoc_ransac_workspace_bytes(_lib.ctx(s.device.index), P, int(seg_src[-1]), int(max_iteration), budget), s.device)
        _lib.check(lib.eyoc_ransac_batched_ws(_lib.ctx(s.device.index), _lib.ptr(s), _lib.ptr(t), _lib.ptr(c), ss, st, P,
                                              C.byref(p), _lib.ptr(res), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()),
                   "eyoc_ransac_batched_ws")
    return res


def _ransac_budget(device) -> int:
    free, _ = torch.cuda.mem_get_info(device)
    cached = torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
    return max(min((free + cached) // 4, 16 << 30), 1)


def decode_ransac_result(res: torch.Tensor, n: int) -> RegistrationResult:
    r = _lib.RansacResult.from_buffer_copy(res.cpu().numpy().tobytes())
    T = np.array(list(r.T), np.float64).reshape(4, 4)
    return RegistrationResult(T, r.inliers / max(n, 1), float(r.inlier_rmse), int(r.inliers),
                              int(r.best_hypothesis), int(r.survivors))


def registration_ransac_based_on_feature_matching(source, target, source_feature, target_feature,
                                                  mutual_filter=False, max_correspondence_distance=0.3,
                                                  estimation_method=None, ransac_n=4, checkers=None,
                                                  criteria=(4000000, 0.999), seed=0, edge_similarity=0.9):
    """Drop-in for the Open3D call of ``scripts/test_kitti.py:171-176``.

    ``source/target``: ``[n,3]`` points (numpy or torch); ``*_feature``: ``[n,C]`` rows (the layout of
    the model output - not Open3D's transposed ``[C,n]``).  ``criteria = (max_iteration, confidence)``:
    like Open3D >= 0.12 with the reference's ``confidence = 10000`` (clamped to 1), every iteration
    runs.  Correspondences are the feature-space nearest neighbours of the source points."""
    if mutual_filter:
        raise NotImplementedError("mutual_filter=True is not used by the reference path")
    if ransac_n != 4:
        raise NotImplementedError("ransac_n must be 4 (scripts/test_kitti.py:174)")
    F0 = _cuda_f32(source_feature)
    F1 = _cuda_f32(target_feature, F0.device)
    idx = knn1_segmented(F0, F1, [0, F0.shape[0]], [0, F1.shape[0]], "SquareL2", return_distance=False)
    return ransac_from_correspondences(torch.as_tensor(np.asarray(source)) if not isinstance(source, torch.Tensor) else source,
                                       torch.as_tensor(np.asarray(target)) if not isinstance(target, torch.Tensor) else target,
                                       idx, max_correspondence_distance, int(criteria[0]), seed, edge_similarity)


class Matcher:
    """scripts/SC2_PCR/SC2_PCR.py:7-31 - same constructor; the KITTI values come from
    scripts/SC2_PCR/config_json/config_KITTI.json."""

    def __init__(self, inlier_threshold=0.10, num_node='all', use_mutual=True, d_thre=0.1, num_iterations=10,
                 ratio=0.2, nms_radius=0.1, max_points=8000, k1=30, k2=20, heatmap=False):
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
        self.heatmap = heatmap

    def _params(self, n):
        # the library takes int(ratio * n) seeds; hand it a ratio that floors to exactly Python's
        # int(num_corr * self.ratio) whatever fp32 does to the literal
        n_seed = int(n * self.ratio)
        return _lib.Sc2pcrParams(float(self.inlier_threshold), float(self.d_thre), (n_seed + 0.5) / n,
                                 float(self.nms_radius), int(self.num_iterations), int(self.max_points),
                                 int(self.k1), int(self.k2)), n_seed

    def match_pair(self, src_keypts, tgt_keypts, src_features, tgt_features, rng=None):
        """:280-305.  ``rng`` replaces the reference's global ``np.random`` for the resampling to
        ``num_node`` (with replacement).  The nearest neighbour is the reference's own quantity,
        ``argmin_j sqrt(2 - 2 <s_i,t_j> + 1e-6)`` (:296-298, ``eyoc_knn1`` dist_type 2): for descriptors that are not
        unit-norm that is an arg-max of the inner product, not an L2 neighbour, and a NaN (inner product above 1)
        wins the row at its first occurrence - both reproduced."""
        rng = np.random if rng is None else rng
        N_src, N_tgt = src_features.shape[1], tgt_features.shape[1]
        if self.num_node == 'all':
            src_sel_ind, tgt_sel_ind = np.arange(N_src), np.arange(N_tgt)
        else:
            src_sel_ind = rng.choice(N_src, self.num_node)
            tgt_sel_ind = rng.choice(N_tgt, self.num_node)
        src_desc = src_features[:, src_sel_ind, :]
        tgt_desc = tgt_features[:, tgt_sel_ind, :]
        src_keypts = src_keypts[:, src_sel_ind, :]
        tgt_keypts = tgt_keypts[:, tgt_sel_ind, :]
        idx = knn1_segmented(src_desc[0], tgt_desc[0], [0, src_desc.shape[1]], [0, tgt_desc.shape[1]],
                             "GemmL2", return_distance=False)
        return src_keypts, tgt_keypts[:, idx.to(tgt_keypts.device)]

    def SC2_PCR(self, src_keypts, tgt_keypts):
        """:307-384 for bs == 1 -> ``(T [1,4,4], seedwise_fitness [1, int(ratio*N)])``."""
        if src_keypts.shape[0] != 1:
            raise NotImplementedError("bs must be 1, as in the reference (SC2_PCR.py:44,249)")
        dev_in = src_keypts.device
        s = _cuda_f32(src_keypts[0])
        t = _cuda_f32(tgt_keypts[0], s.device)
        n = min(t.shape[0], int(self.max_points))
        s, t = s[:n].contiguous(), t[:n].contiguous()
        lib = _lib.load()
        p, n_seed = self._params(n)
        T = torch.empty((1, 4, 4), dtype=torch.float32, device=s.device)
        fit = torch.zeros((1, max(n_seed, 1)), dtype=torch.float32, device=s.device)
        with torch.cuda.device(s.device):
            ws = _lib.workspace(lib.eyoc_sc2pcr_workspace_bytes(n, C.byref(p)), s.device)
            _lib.check(lib.eyoc_sc2pcr(_lib.ctx(s.device.index), _lib.ptr(s), _lib.ptr(t), n, C.byref(p), _lib.ptr(T),
                                       _lib.ptr(fit), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "eyoc_sc2pcr")
        return T.to(dev_in), fit[:, :n_seed].to(dev_in)

    def SC2_PCR_packed(self, src, tgt, seg):
        """Batched ``SC2_PCR`` on packed inputs: ``src / tgt f32 [N,3]`` hold the matched key points of all pairs back
        to back, pair ``b`` = rows ``seg[b]:seg[b+1]`` (each at most ``max_points`` rows: truncate before packing).
        Returns ``(T [B,4,4], fitness [B, stride], n_seed list)`` on the device; every pair is bit-identical to
        ``SC2_PCR`` on it alone.  16 pairs share a launch (``eyoc_sc2pcr_batched``)."""
        src = _cuda_f32(src)
        tgt = _cuda_f32(tgt, src.device)
        dev = src.device
        seg = [int(v) for v in seg]
        ns = [b - a for a, b in zip(seg[:-1], seg[1:])]
        if max(ns) > int(self.max_points):
            raise ValueError("a pair exceeds max_points: truncate before packing")
        B = len(ns)
        segc = (C.c_int32 * (B + 1))(*seg)
        pn = [self._params(n) for n in ns]
        params = (_lib.Sc2pcrParams * B)(*[p for p, _ in pn])
        stride = max(max(k for _, k in pn), 1)
        T = torch.empty((B, 4, 4), dtype=torch.float32, device=dev)
        fit = torch.zeros((B, stride), dtype=torch.float32, device=dev)
        lib = _lib.load()
        i_max = int(np.argmax(ns))
        with torch.cuda.device(dev):
            ws = _lib.workspace(lib.eyoc_sc2pcr_batched_workspace_bytes_n(ns[i_max], B, C.byref(params[i_max])), dev)
            _lib.check(lib.eyoc_sc2pcr_batched(_lib.ctx(dev.index), _lib.ptr(src), _lib.ptr(tgt), segc, B, params, _lib.ptr(T),
                                               _lib.ptr(fit), stride, _lib.ptr(ws), ws.numel(), _lib.stream_ptr()),
                       "eyoc_sc2pcr_batched")
        return T, fit, [k for _, k in pn]

    def SC2_PCR_batch(self, src_list, tgt_list):
        """The loop of lib/trainer.py:1157-1166 (``SC2_PCR`` once per pair, which the reference leaves sequential)
        as one batched call: lists of ``[n_b,3]`` matched key points -> list of ``(T [4,4], seedwise_fitness)``,
        each bit-identical to ``SC2_PCR(src[None], tgt[None])`` on that pair."""
        ss = [_cuda_f32(s) for s in src_list]
        dev = ss[0].device
        ns = [min(s.shape[0], int(self.max_points)) for s in ss]
        src = torch.cat([s[:n] for s, n in zip(ss, ns)]).contiguous()
        tgt = torch.cat([_cuda_f32(t, dev)[:n] for t, n in zip(tgt_list, ns)]).contiguous()
        T, fit, n_seed = self.SC2_PCR_packed(src, tgt, np.concatenate([[0], np.cumsum(ns)]))
        return [(T[b], fit[b, :n_seed[b]]) for b in range(len(ns))]

    def estimator(self, src_keypts, tgt_keypts, src_features, tgt_features, rng=None):
        """:386-413 -> ``(T, labels, src_corr, tgt_corr, seedwise_fitness)``."""
        src_corr, tgt_corr = self.match_pair(src_keypts, tgt_keypts, src_features, tgt_features, rng)
        pred_trans, fitness = self.SC2_PCR(src_corr, tgt_corr)
        warp = transform(src_corr, pred_trans.to(src_corr.device))
        distance = torch.sum((warp - tgt_corr) ** 2, dim=-1) ** 0.5
        pred_labels = (distance < self.inlier_threshold).float()
        return pred_trans, pred_labels, src_corr, tgt_corr, fitness
