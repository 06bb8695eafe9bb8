"""Deterministic inputs shared by ``make_golden.py`` (which runs the reference on them) and the
tests (which run the oracle / the HIP path on them).  Only PCG64 uniform doubles, element-wise IEEE
operations and explicit left-to-right accumulation are used, so the arrays are bit-reproducible on
any machine and the fixtures only have to store the reference's OUTPUTS."""
import numpy as np


def _u(seed, *shape):
    return np.random.Generator(np.random.PCG64(seed)).random(shape)


def _normal(seed, *shape):
    u = _u(seed, 2, *shape)
    return np.sqrt(-2.0 * np.log(1.0 - u[0])) * np.cos(2.0 * np.pi * u[1])


def _rownorm(x):
    acc = np.zeros(x.shape[0])
    for c in range(x.shape[1]):
        acc = acc + x[:, c] * x[:, c]
    return x / np.sqrt(acc)[:, None]


def unit_feats(seed, n, c=32):
    return _rownorm(_normal(seed, n, c)).astype(np.float32)


def nn_case(seed, n0, n1, c=32, noise=0.35):
    """F0 random unit rows; F1 = noisy copies of a random subset of F0 plus distractors."""
    F0 = _normal(seed, n0, c)
    src = (_u(seed + 1, n1) * n0).astype(np.int64)
    F1 = F0[src] + noise * _normal(seed + 2, n1, c)
    return _rownorm(F0).astype(np.float32), _rownorm(F1).astype(np.float32)


def loss_case(seed, n0, n1, n_pairs, c=32, noise=0.25):
    """Two descriptor sets and ``positive_pairs [n_pairs, 2]`` for the hardest-contrastive loss: the partner of a positive
    is a noisy copy (so positives are close, some beyond the positive margin), a tenth of the pairs is listed twice, and
    the candidate sets hold near-duplicates of anchors (hard negatives, some of them known positives: the mask matters)."""
    F0 = _normal(seed, n0, c)
    i0 = (_u(seed + 1, n_pairs) * n0).astype(np.int64)
    i1 = (_u(seed + 2, n_pairs) * n1).astype(np.int64)
    dup = np.nonzero(_u(seed + 3, n_pairs) < 0.1)[0]
    i0[dup] = i0[(dup * 7) % n_pairs]
    i1[dup] = i1[(dup * 7) % n_pairs]
    F1 = _normal(seed + 4, n1, c)
    F1[i1] = F0[i0] + noise * _normal(seed + 5, n_pairs, c)      # later pairs overwrite earlier ones: some positives end up far apart
    return _rownorm(F0).astype(np.float32), _rownorm(F1).astype(np.float32), np.stack([i0, i1], 1)


def rot_zyx(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def rigid(rx, ry, rz, tx, ty, tz):
    T = np.eye(4)
    T[:3, :3] = rot_zyx(rx, ry, rz)
    T[:3, 3] = (tx, ty, tz)
    return T


def corr_case(seed, n, T, inlier_frac, noise=0.02, extent=(60.0, 60.0, 6.0)):
    """Correspondences ``p1 ~ T p0`` with a fraction of gross outliers (uniform in the scene box)."""
    p0 = (_u(seed, n, 3) - 0.5) * np.array(extent)
    p1 = p0 @ T[:3, :3].T + T[:3, 3] + noise * _normal(seed + 1, n, 3)
    out = _u(seed + 2, n) >= inlier_frac
    p1[out] = (_u(seed + 3, n, 3)[out] - 0.5) * np.array(extent)
    return p0.astype(np.float32), p1.astype(np.float32), ~out


def match_pair_case(kind, seed, n0, n1, c=32):
    """Descriptors for ``Matcher.match_pair``: ``unit`` = noisy unit-norm copies (``nn_case``); ``ties`` = unit-norm with
    every target present three times (exact ties: the lowest index must win) and 200 targets one ulp apart; ``raw`` =
    rows of norm 0.5 .. 1.6, NOT normalised (inner products above 1 make the reference's sqrt NaN for some rows)."""
    if kind == "unit":
        return nn_case(seed, n0, n1, c)
    if kind == "ties":
        F0, F1 = nn_case(seed, n0, n1 // 3, c)
        F1 = np.concatenate([F1, F1, F1])[:n1].copy()
        k = min(200, len(F1))
        F1[:k, 0] = np.nextafter(F1[:k, 0], np.float32(2.0))
        return F0, F1
    if kind == "raw":
        F0, F1 = nn_case(seed, n0, n1, c)
        s0 = (0.5 + 1.1 * _u(seed + 5, n0)).astype(np.float32)
        s1 = (0.5 + 1.1 * _u(seed + 6, n1)).astype(np.float32)
        return F0 * s0[:, None], F1 * s1[:, None]
    raise ValueError(kind)


def label_batch_case(seed, sizes=((1500, 1300), (800, 2000), (1000, 1000))):
    """A batch of cloud pairs for ``match_and_filter_corr`` (G10): coordinates around two sensors (norms on both sides of the 20 m
    spherical radius and across the cells of the similarity tables), unit features of which half are shared with noise."""
    C0s, F0s, C1s, F1s = [], [], [], []
    for k, (n0, n1) in enumerate(sizes):
        C0s.append(((_u(seed + 10 * k, n0, 3) - 0.5) * 120).astype(np.float32))
        C1s.append(((_u(seed + 10 * k + 1, n1, 3) - 0.5) * 120).astype(np.float32))
        F0, F1 = nn_case(seed + 10 * k + 2, n0, n1, noise=0.15)
        F0s.append(F0); F1s.append(F1)
    return C0s, F0s, C1s, F1s


def dist_sim_table(seed=0):
    """A similarity table of the reference's shape family (config/dist_sim_plot/*.npz: six float64 slices [gap cells, distance
    cells]) - synthetic data; the reference's own tables stay with the reference."""
    rng = np.random.default_rng(seed)
    out = {}
    for i, shape in enumerate([(12, 16), (18, 16), (20, 18), (20, 18), (20, 18), (20, 18)]):
        gy, gx = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing="ij")
        out[i] = np.clip(0.85 * np.exp(-0.12 * gy - 0.05 * gx) + 0.05 * rng.normal(size=shape), 0.0, 1.0)
    return out
