"""CPU oracle of the feature-matching glue (numpy).  TEST INFRASTRUCTURE.

Pinned against golden vectors generated from the reference's own ``lib.metrics.pdist`` /
``lib.eval.find_nn_gpu`` (``tests/golden/make_golden.py`` -> ``tests/golden/g1_nn.npz``).

Restated functions:
  * ``pdist``           lib/metrics.py:22-29
  * ``find_nn_gpu``     lib/eval.py:18-48
  * ``find_corr``       scripts/test_kitti.py:28-42 (twin at lib/trainer.py:405-419)
  * ``random_sample``   scripts/test_kitti.py:54-73
  * ``match_pair``      scripts/SC2_PCR/SC2_PCR.py:280-305 (GEMM-form nearest neighbour; pinned by golden g6 from the
                        reference's own ``Matcher.match_pair``)

Arithmetic contract shared with the HIP kernel (``eyoc_knn1``), chosen so indices are bit-exact
between the two: per (i, j) the squared distance is accumulated in fp32, channels left to right,
each term ``d = a - b; s = d * d; acc = acc + s`` rounded separately (no FMA); the minimum keeps
the lowest j among equal distances.  The reference's ``torch.sum`` reduces the same 32 terms in an
implementation-defined order, so its indices can differ from this only where two candidates are
within a few ulp - the golden test audits exactly those rows.
"""
from __future__ import annotations

import numpy as np


def sqdist_rows(A: np.ndarray, B: np.ndarray, out_dtype=np.float32) -> np.ndarray:
    """``D[i,j] = sum_c (A[i,c]-B[j,c])^2`` with the sequential fp32 contract above."""
    A = np.ascontiguousarray(A, np.float32)
    B = np.ascontiguousarray(B, np.float32)
    n, c = A.shape
    acc = np.zeros((n, B.shape[0]), np.float32)
    for ch in range(c):
        d = A[:, ch][:, None] - B[:, ch][None, :]
        acc = acc + d * d
    return acc.astype(out_dtype, copy=False)


def pdist(A, B, dist_type="L2"):
    """lib/metrics.py:22-29."""
    D2 = sqdist_rows(A, B)
    if dist_type == "L2":
        return np.sqrt(D2 + np.float32(1e-7))
    if dist_type == "SquareL2":
        return D2
    raise NotImplementedError("Not implemented")


def find_nn(F0, F1, nn_max_n=-1, return_distance=False, dist_type="SquareL2", chunk=256):
    """lib/eval.py:18-48 - index (int64) of the nearest row of F1 for every row of F0.

    ``nn_max_n`` only changes the reference's memory chunking, never its result, so it is accepted
    and ignored; ``chunk`` bounds this oracle's own memory."""
    F0 = np.ascontiguousarray(F0, np.float32)
    inds = np.empty(len(F0), np.int64)
    dists = np.empty((len(F0), 1), np.float32)
    for s in range(0, len(F0), chunk):
        D = pdist(F0[s:s + chunk], F1, dist_type)
        j = np.argmin(D, axis=1)                 # first minimum == lowest index
        inds[s:s + chunk] = j
        dists[s:s + chunk, 0] = D[np.arange(len(j)), j]
    return (inds, dists) if return_distance else inds


def find_corr(xyz0, xyz1, F0, F1, subsample_size=-1, inds0=None, inds1=None, rng=None):
    """scripts/test_kitti.py:28-42.  The reference draws the sub-sample with the global
    ``np.random``; here the draw is injectable (``inds0/inds1``) so both sides of a parity test
    use identical indices."""
    subsample = len(F0) > subsample_size
    if subsample_size > 0 and subsample:
        if inds0 is None:
            rng = rng or np.random.default_rng(0)
            inds0 = rng.choice(len(F0), min(len(F0), subsample_size), replace=False)
            inds1 = rng.choice(len(F1), min(len(F1), subsample_size), replace=False)
        F0, F1 = F0[inds0], F1[inds1]
    nn = find_nn(F0, F1, nn_max_n=500)
    if subsample_size > 0 and subsample:
        return xyz0[inds0], xyz1[inds1[nn]]
    return xyz0, xyz1[nn]


def random_sample(pcd, feats, N, rng):
    """scripts/test_kitti.py:54-73 - exactly-N sampling."""
    n1 = pcd.shape[0]
    if n1 == N:
        return pcd, feats
    choice = rng.permutation(n1)[:N] if n1 > N else rng.choice(n1, N)
    return pcd[choice], feats[choice]


def fmaf(a, b, c):
    """Correctly rounded fp32 ``a * b + c`` on arrays (what ``v_fma_f32`` / C ``fmaf`` return), without an fma in
    numpy: the product of two fp32 numbers is exact in fp64; the fp64 sum with ``c`` is made round-to-odd (TwoSum gives
    the rounding error's sign; an inexact even result moves one ulp towards it), and rounding an odd-rounded fp64 value
    to fp32 equals rounding the exact value (29 guard bits)."""
    p = np.asarray(a, np.float32).astype(np.float64) * np.asarray(b, np.float32).astype(np.float64)
    c = np.broadcast_to(np.asarray(c, np.float32).astype(np.float64), p.shape)
    with np.errstate(invalid="ignore", over="ignore"):
        s = p + c
        bb = s - p
        err = (p - (s - bb)) + (c - bb)
        fix = (err != 0) & np.isfinite(s) & ((s.view(np.int64) & 1) == 0)
        s = np.where(fix, np.nextafter(s, np.where(err > 0, np.inf, -np.inf)), s)
        return s.astype(np.float32)


def dot_rows(A, B):
    """``S[i,j] = <A_i, B_j>`` as the fp32 FMA chain over the channels in order, starting from 0 - the arithmetic
    contract of ``eyoc_dotmax`` and of ``eyoc_knn1`` dist_type 2."""
    A = np.ascontiguousarray(A, np.float32)
    B = np.ascontiguousarray(B, np.float32)
    acc = np.zeros((A.shape[0], B.shape[0]), np.float32)
    for ch in range(A.shape[1]):
        acc = fmaf(A[:, ch][:, None], B[:, ch][None, :], acc)
    return acc


def match_pair_distance(src_desc, tgt_desc):
    """scripts/SC2_PCR/SC2_PCR.py:296: ``sqrt(2 - 2 * (src @ tgt.T) + 1e-6)`` in fp32, each step rounded (the
    reference's sgemm sums the 32 products in an implementation-defined order; here, and in the HIP kernel, it is
    ``dot_rows``' chain - results can differ from the reference only where two candidates are within rounding)."""
    S = dot_rows(src_desc, tgt_desc)
    with np.errstate(invalid="ignore"):
        return np.sqrt((np.float32(2) - np.float32(2) * S) + np.float32(1e-6))


def match_pair_indices(src_desc, tgt_desc, chunk=512):
    """scripts/SC2_PCR/SC2_PCR.py:296-298: ``argmin_j sqrt(2 - 2 <s_i, t_j> + 1e-6)`` with ``torch.argmin``'s rules:
    the first NaN wins its row (inner products above 1 + 5e-7 - descriptors that are not unit-norm), otherwise the
    first minimum.  (``numpy.argmin`` treats NaN the same way.)"""
    src_desc = np.ascontiguousarray(src_desc, np.float32)
    out = np.empty(len(src_desc), np.int64)
    for s in range(0, len(src_desc), chunk):
        out[s:s + chunk] = np.argmin(match_pair_distance(src_desc[s:s + chunk], tgt_desc), axis=1)
    return out
