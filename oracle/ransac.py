"""CPU oracle of feature-matching RANSAC (numpy, fp64).  TEST INFRASTRUCTURE.

**Parity unpinned.**  The reference calls Open3D (requirements.txt:5, un-pinned, not installable
here): ``o3d.pipelines.registration.registration_ransac_based_on_feature_matching(pcd0, pcd1,
feat0, feat1, False, voxel_size, TransformationEstimationPointToPoint(False), 4,
[CorrespondenceCheckerBasedOnEdgeLength(0.9), CorrespondenceCheckerBasedOnDistance(voxel_size)],
RANSACConvergenceCriteria(4000000, 10000))`` at scripts/test_kitti.py:169-177.  Open3D's sampler is
seeded from ``std::random_device`` and runs under OpenMP, so its result is not reproducible even
against itself; this file restates the published algorithm (Open3D >= 0.12) with an explicit
counter-based sampler so that the HIP kernel can be checked hypothesis by hypothesis:

  1. correspondences: every source feature -> its nearest target feature (``mutual_filter=False``);
  2. per hypothesis h: draw 4 correspondence indices ``sample(seed, h, t) t=0..3``;
  3. edge-length checker on the 6 point pairs of the sample:
     reject if ``|s_a - s_b| < 0.9 |t_a - t_b|`` or ``|t_a - t_b| < 0.9 |s_a - s_b|`` (evaluated on the
     squared lengths, which is the same predicate without square roots);
  4. rigid transform of the 4 pairs (Kabsch / ``Eigen::umeyama`` without scale);
  5. distance checker: reject unless all 4 residuals ``|T s - t| <= max_distance``;
  6. score on ALL correspondences: inliers = ``|T s - t| < max_distance`` (evaluated as
     ``|T s - t|^2 < max_distance^2`` - the same predicate without a square root per residual); fitness = inliers / n,
     inlier RMSE; keep the hypothesis with higher fitness, then lower RMSE, then lower h;
  7. ``RANSACConvergenceCriteria(4000000, 10000)``: the confidence argument is clamped to 1, which
     disables early termination - all ``max_iteration`` hypotheses are evaluated.

Sampler (shared bit-for-bit with ``eyoc_amd/csrc/ransac.hip``): splitmix64 finaliser of the counters
``seed * 0x9E3779B97F4A7C15 + 2 h + t`` (t = 0, 1); each 64-bit word yields two indices, its low and its high
32 bits ``u`` mapped to ``[0, n)`` by ``(u * n) >> 32`` - samples 0, 1 from word 0 and 2, 3 from word 1.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def sample_indices(seed: int, h0: int, count: int, n: int) -> np.ndarray:
    """``[count, 4] int64`` correspondence indices for hypotheses ``h0 .. h0+count-1``."""
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
        ctr = (np.arange(h0, h0 + count, dtype=np.uint64)[:, None] * np.uint64(2)
               + np.arange(2, dtype=np.uint64)[None, :]) + base
        x = ctr
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
        u = np.stack([x & np.uint64(0xFFFFFFFF), x >> np.uint64(32)], axis=2).reshape(count, 4)
        return ((u * np.uint64(n)) >> np.uint64(32)).astype(np.int64)


def kabsch(S: np.ndarray, T: np.ndarray) -> np.ndarray:
    """Batched un-weighted rigid fit ``[B,n,3] -> [B,4,4]`` with ``T ~ R S + t`` (fp64)."""
    cs, ct = S.mean(1, keepdims=True), T.mean(1, keepdims=True)
    H = np.einsum("bni,bnj->bij", S - cs, T - ct)
    U, _, Vt = np.linalg.svd(H)
    V = np.swapaxes(Vt, 1, 2)
    d = np.linalg.det(V @ np.swapaxes(U, 1, 2))
    D = np.tile(np.eye(3), (len(S), 1, 1))
    D[:, 2, 2] = d
    R = V @ D @ np.swapaxes(U, 1, 2)
    t = ct[:, 0, :] - np.einsum("bij,bj->bi", R, cs[:, 0, :])
    out = np.tile(np.eye(4), (len(S), 1, 1))
    out[:, :3, :3] = R
    out[:, :3, 3] = t
    return out


def ransac(src: np.ndarray, tgt: np.ndarray, corr_tgt: np.ndarray, max_distance: float,
           max_iteration: int, seed: int = 0, edge_similarity: float = 0.9, chunk: int = 1 << 20):
    """``src f[n,3]``, ``tgt f[m,3]``, ``corr_tgt int[n]`` (target index of source point i).

    Returns ``dict(T f64[4,4], inliers int, rmse float, best_h int, survivors int)``.
    """
    # the C ABI carries both thresholds as fp32 (eyoc_ransac_params); mirror that rounding
    max_distance = float(np.float32(max_distance))
    edge_similarity = float(np.float32(edge_similarity))
    S_all = np.asarray(src, np.float64)
    T_all = np.asarray(tgt, np.float64)[np.asarray(corr_tgt)]
    n = len(S_all)
    best = (-1, np.inf, -1, np.eye(4))
    survivors = 0
    pairs = [(a, b) for a in range(4) for b in range(a + 1, 4)]
    for h0 in range(0, max_iteration, chunk):
        cnt = min(chunk, max_iteration - h0)
        idx = sample_indices(seed, h0, cnt, n)
        s, t = S_all[idx], T_all[idx]                                    # [cnt,4,3]
        ok = np.ones(cnt, bool)
        e2 = edge_similarity * edge_similarity
        for a, b in pairs:   # squared form of the edge-length checker, as the kernel evaluates it
            ds2 = ((s[:, a] - s[:, b]) ** 2).sum(1)
            dt2 = ((t[:, a] - t[:, b]) ** 2).sum(1)
            ok &= ~((ds2 < dt2 * e2) | (dt2 < ds2 * e2))
        cand = np.nonzero(ok)[0]
        if len(cand) == 0:
            continue
        Ts = kabsch(s[cand], t[cand])
        res = np.linalg.norm(np.einsum("bij,bnj->bni", Ts[:, :3, :3], s[cand]) + Ts[:, None, :3, 3]
                             - t[cand], axis=2)
        keep = ~(res > max_distance).any(1)
        cand, Ts = cand[keep], Ts[keep]
        survivors += len(cand)
        for c0 in range(0, len(cand), 256):
            Tc = Ts[c0:c0 + 256]
            r = np.einsum("bij,nj->bni", Tc[:, :3, :3], S_all) + Tc[:, None, :3, 3] - T_all[None]
            d2 = (r * r).sum(2)
            inl = d2 < max_distance * max_distance     # squared form of `dist < max_distance`, as the kernel evaluates it
            cntc = inl.sum(1)
            err2 = np.where(inl, d2, 0.0).sum(1)
            with np.errstate(invalid="ignore", divide="ignore"):
                rmse = np.where(cntc > 0, np.sqrt(err2 / np.maximum(cntc, 1)), np.inf)
            rmse = rmse.astype(np.float32).astype(np.float64)   # the device ranks by the fp32 RMSE
            for j in range(len(Tc)):
                key = (int(cntc[j]), float(rmse[j]), int(h0 + cand[c0 + j]))
                if key[0] > best[0] or (key[0] == best[0] and (key[1] < best[1] or
                                        (key[1] == best[1] and key[2] < best[2]))):
                    best = (key[0], key[1], key[2], Tc[j])
    return {"T": best[3], "inliers": best[0], "rmse": best[1], "best_h": best[2],
            "survivors": survivors, "fitness": max(best[0], 0) / max(n, 1)}
