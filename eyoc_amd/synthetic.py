"""Deterministic KITTI-shaped synthetic input for the registration hot path.

There is no dataset and no checkpoint offline, so parity tests and ``bench.py`` run on a seeded
ray-cast street scene: a 64-beam / 2000-azimuth / 80 m sensor looks at a ground
plane, facades, boxes, poles and vegetation blobs; the second cloud of a pair is the same scene seen
from a pose displaced by ``d ~ U[5,20] m`` with a small yaw, so the ground-truth transform is known.

The voxelisation contract follows the reference's loader (``lib/data_loaders.py:940-979``):
``sel = first point of every occupied voxel of floor(xyz / voxel)`` in input order,
``coords = floor(xyz[sel] / voxel).int()``, features are all ones, and the batch index is prepended
as column 0 (``lib/data_loaders.py:65-66`` via ``ME.utils.sparse_collate``).

Only numpy is used here; nothing in this file touches the GPU or the oracle.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "make_scene", "raycast", "voxelize", "make_pair", "make_weights", "subsample_indices",
    "batch_coords", "RESUNET_BN2C_LAYOUT", "plant_correspondences",
]

GROUND_Z = 0.0
SENSOR_H = 1.73


def _rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def make_scene(rng, extent=110.0):
    """Static street scene in world coordinates (x = driving direction)."""
    sc = {}
    boxes = []
    # facades: long oriented boxes on both sides of the street, broken into segments
    x = -extent
    while x < extent + 40:
        for side in (-1.0, 1.0):
            if rng.random() < 0.7:
                ln = rng.uniform(8.0, 25.0)
                depth = rng.uniform(6.0, 12.0)
                h = rng.uniform(4.0, 14.0)
                off = rng.uniform(14.0, 32.0)
                cy = side * (off + depth / 2)
                boxes.append((x + ln / 2, cy, h / 2, ln / 2, depth / 2, h / 2, rng.uniform(-0.06, 0.06)))
        x += rng.uniform(14.0, 30.0)
    # free-standing boxes (cars, containers, kiosks): 1.5 - 12 m
    for _ in range(60):
        sx = rng.uniform(1.5, 12.0) / 2
        sy = rng.uniform(1.5, 5.0) / 2
        sz = rng.uniform(1.2, 4.0) / 2
        cx = rng.uniform(-extent, extent + 30)
        cy = rng.choice([-1.0, 1.0]) * rng.uniform(3.5, 30.0)
        boxes.append((cx, cy, sz, sx, sy, sz, rng.uniform(0, np.pi)))
    sc["boxes"] = np.array(boxes)
    # poles / trunks
    npole = 40
    sc["poles"] = np.stack([
        rng.uniform(-extent, extent + 30, npole),
        rng.choice([-1.0, 1.0], npole) * rng.uniform(4.0, 25.0, npole),
        rng.uniform(0.08, 0.25, npole),
        rng.uniform(3.0, 9.0, npole)], 1)
    # vegetation: participating-media boxes (hedges, bushes, tree canopies); a ray that enters one
    # returns from an exponentially distributed depth, which is what makes real sweeps voxel-rich
    veg = []
    nveg = 35
    for _ in range(nveg):
        hx = rng.uniform(2.0, 14.0)
        hy = rng.uniform(1.5, 7.0)
        top = rng.uniform(1.5, 9.0)
        bot = 0.0 if rng.random() < 0.6 else rng.uniform(1.5, 3.0)  # bush vs canopy
        cx = rng.uniform(-extent, extent + 30)
        cy = rng.choice([-1.0, 1.0]) * rng.uniform(5.5 + hy, 42.0)
        veg.append((cx, cy, 0.5 * (top + bot), hx, hy, 0.5 * (top - bot), rng.uniform(0, np.pi),
                    rng.uniform(1.0, 4.0)))
    # far backdrop: tree lines / building fronts 40-75 m out, so the upper beams return at all
    for _ in range(70):
        ang = rng.uniform(0, 2 * np.pi)
        r = rng.uniform(40.0, 75.0)
        cx, cy = 10.0 + 1.3 * r * np.cos(ang), r * np.sin(ang)
        h = rng.uniform(8.0, 20.0)
        veg.append((cx, cy, h / 2, rng.uniform(5.0, 12.5), rng.uniform(3.0, 8.0), h / 2,
                    ang + np.pi / 2 + rng.uniform(-0.3, 0.3), rng.uniform(1.0, 4.0)))
    sc["veg"] = np.array(veg).reshape(-1, 8)
    sc["rough"] = rng.uniform(0, 2 * np.pi, 6)
    return sc


def _brush_boxes(level):
    """Off-road brush layer (grass, shrubs): two huge thin media boxes, one per road side. ``level``
    is the only knob ``make_pair`` bisects: it sets the layer height and mean free path."""
    h = 0.25 + 0.9 * level
    mfp = 3.0 + 3.5 * level
    return [(10.0, side * 51.0, h / 2, 170.0, 45.0, h / 2, 0.0, mfp) for side in (-1.0, 1.0)]


def _terrain(ph, x, y):
    """Gentle undulation + curbs; keeps the road itself near z = 0."""
    h = 0.22 * np.sin(0.05 * x + ph[0]) * np.cos(0.04 * y + ph[1])
    h += 0.10 * np.sin(0.21 * x + ph[2]) + 0.08 * np.sin(0.33 * y + ph[3])
    h += 0.03 * np.sin(0.7 * x + ph[4]) * np.cos(0.9 * y + ph[5])  # +-3 cm roughness
    h += 0.15 * (np.abs(y) > 4.8)  # curb
    return h


def _ray_dirs(beams, azimuths, elev_top=2.0, elev_bot=-24.8):
    """Ring-major ray directions. 64 beams use the HDL-64E split (upper 32 lasers at 1/3 deg from
    +2 deg, lower 32 at 1/2 deg down to -24.8 deg); other beam counts are spaced uniformly."""
    if beams == 64 and elev_top == 2.0 and elev_bot == -24.8:
        el = np.concatenate([2.0 - np.arange(32) / 3.0, -8.83 - np.arange(32) * (15.97 / 31.0)])
    else:
        el = np.linspace(elev_top, elev_bot, beams)
    el = np.deg2rad(el)
    az = np.linspace(-np.pi, np.pi, azimuths, endpoint=False)
    ce, se = np.cos(el)[:, None], np.sin(el)[:, None]
    d = np.stack([ce * np.cos(az)[None, :], ce * np.sin(az)[None, :], se * np.ones_like(az)[None, :]], -1)
    return d.reshape(-1, 3)  # ring-major, like a KITTI .bin


def _slab(o, d, box):
    cx, cy, cz, hx, hy, hz, yaw = box[:7]
    Rb = _rot_z(-yaw)
    ob = Rb @ (o - np.array([cx, cy, cz]))
    db = d @ Rb.T
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / db
        h = np.array([hx, hy, hz])
        t1 = (-h - ob) * inv
        t2 = (h - ob) * inv
    tn = np.nanmax(np.minimum(t1, t2), axis=1)
    tf = np.nanmin(np.maximum(t1, t2), axis=1)
    hit = (tf >= tn) & (tf > 0)
    return tn, tf, hit


def _az_slice(o, yaw, box, azimuths):
    """Azimuth index window (in the sensor's ring order) that can see ``box``; None = all rays."""
    cx, cy, hx, hy = box[0], box[1], box[3], box[4]
    rad = np.hypot(hx, hy)
    dist = np.hypot(cx - o[0], cy - o[1])
    if dist <= rad * 1.05:
        return None
    half = np.arcsin(min(1.0, rad / dist)) + 2 * np.pi / azimuths
    mid = np.arctan2(cy - o[1], cx - o[0]) - yaw
    a0 = int(np.floor((mid - half + np.pi) / (2 * np.pi) * azimuths))
    a1 = int(np.ceil((mid + half + np.pi) / (2 * np.pi) * azimuths)) + 1
    if a1 - a0 >= azimuths:
        return None
    return np.arange(a0, a1) % azimuths


def _cast_static(scene, pose, beams, azimuths, max_range, elev_top, elev_bot, rng):
    """Nearest hit distance per ray against everything except the brush layer."""
    d_s = _ray_dirs(beams, azimuths, elev_top, elev_bot)
    R, o = pose[:3, :3], pose[:3, 3]
    yaw = np.arctan2(R[1, 0], R[0, 0])
    d = d_s @ R.T
    n = d.shape[0]
    ph = scene["rough"]
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = (GROUND_Z - o[2]) / d[:, 2]
        for _ in range(2):
            hx = _terrain(ph, o[0] + tg * d[:, 0], o[1] + tg * d[:, 1])
            tg = (hx - o[2]) / d[:, 2]
    tg[(d[:, 2] >= 0) | ~np.isfinite(tg) | (tg <= 0)] = np.inf
    t_best = tg
    ring = np.arange(beams)[:, None] * azimuths

    def rays_for(box):
        az = _az_slice(o, yaw, box, azimuths)
        return None if az is None else (ring + az[None, :]).reshape(-1)

    for box in scene["boxes"]:
        if np.hypot(box[0] - o[0], box[1] - o[1]) - np.hypot(box[3], box[4]) > max_range:
            continue
        idx = rays_for(box)
        dd = d if idx is None else d[idx]
        tn, tf, hit = _slab(o, dd, box)
        t = np.where(hit & (tn > 0), tn, np.inf)
        if idx is None:
            t_best = np.minimum(t_best, t)
        else:
            t_best[idx] = np.minimum(t_best[idx], t)

    for cx, cy, r, h in scene["poles"]:
        ox, oy = o[0] - cx, o[1] - cy
        if np.hypot(ox, oy) > max_range:
            continue
        idx = rays_for((cx, cy, 0, r, r, 0))
        dd = d if idx is None else d[idx]
        a = dd[:, 0] ** 2 + dd[:, 1] ** 2
        b = 2 * (ox * dd[:, 0] + oy * dd[:, 1])
        c = ox * ox + oy * oy - r * r
        disc = b * b - 4 * a * c
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (-b - np.sqrt(np.where(disc > 0, disc, np.nan))) / (2 * a)
        z = o[2] + t * dd[:, 2]
        ok = (disc > 0) & (t > 0) & (z >= GROUND_Z) & (z <= h)
        t = np.where(ok, t, np.inf)
        if idx is None:
            t_best = np.minimum(t_best, t)
        else:
            t_best[idx] = np.minimum(t_best[idx], t)

    for box in scene["veg"]:
        if np.hypot(box[0] - o[0], box[1] - o[1]) - np.hypot(box[3], box[4]) > max_range:
            continue
        idx = rays_for(box)
        dd = d if idx is None else d[idx]
        tn, tf, hit = _slab(o, dd, box)
        t = np.maximum(tn, 0.0) + rng.exponential(box[7], len(dd))
        t = np.where(hit & (t < tf), t, np.inf)
        if idx is None:
            t_best = np.minimum(t_best, t)
        else:
            t_best[idx] = np.minimum(t_best[idx], t)
    return d, t_best


def _finish(pose, d, t_static, brush_level, rng, max_range, noise):
    R, o = pose[:3, :3], pose[:3, 3]
    t_best = t_static
    for box in _brush_boxes(brush_level):
        tn, tf, hit = _slab(o, d, box)
        t = np.maximum(tn, 0.0) + rng.exponential(box[7], len(d))
        t_best = np.minimum(t_best, np.where(hit & (t < tf), t, np.inf))
    ok = np.isfinite(t_best) & (t_best < max_range) & (t_best > 1.5)
    pts_w = o[None, :] + t_best[ok][:, None] * d[ok]
    pts_s = (pts_w - o[None, :]) @ R  # world -> sensor (R^T applied on the right)
    pts_s += rng.normal(0.0, noise, pts_s.shape)
    return pts_s.astype(np.float32)


def raycast(scene, pose, rng, brush_level=1.0, beams=64, azimuths=2000, max_range=80.0, noise=0.02,
            elev_top=2.0, elev_bot=-24.8):
    """Return hit points ``float32 [M,3]`` in the SENSOR frame of ``pose`` (4x4 sensor->world)."""
    d, t = _cast_static(scene, pose, beams, azimuths, max_range, elev_top, elev_bot, rng)
    return _finish(pose, d, t, brush_level, rng, max_range, noise)


def voxelize(xyz, voxel_size):
    """First-point-per-voxel selection in input order + integer coordinates.

    Mirrors ``ME.utils.sparse_quantize(xyz / voxel, return_index=True)`` followed by
    ``floor(xyz[sel] / voxel).int()`` (``lib/data_loaders.py:940-943,969-979``); the selection is
    returned sorted so that the voxelised cloud keeps the sensor's point order.
    """
    c = np.floor(xyz.astype(np.float32) / np.float32(voxel_size)).astype(np.int64)
    key = ((c[:, 0] + (1 << 20)) << 42) | ((c[:, 1] + (1 << 20)) << 21) | (c[:, 2] + (1 << 20))
    _, first = np.unique(key, return_index=True)
    sel = np.sort(first)
    return sel, c[sel].astype(np.int32)


def _pose(x, y, yaw):
    T = np.eye(4)
    T[:3, :3] = _rot_z(yaw)
    T[:3, 3] = (x, y, SENSOR_H)
    return T


def make_pair(seed, voxel_size=0.3, dist_range=(5.0, 20.0), beams=64, azimuths=2000,
              band=(28500, 31500), max_yaw_deg=10.0, max_range=80.0, elev=(2.0, -24.8)):
    """One synthetic registration pair.

    Returns a dict with ``xyz0/xyz1 f32 [N,3]`` (voxelised points), ``coords0/coords1 i32 [N,3]``,
    ``feats0/feats1 f32 [N,1]`` (ones) and ``T_gt f32 [4,4]`` with ``xyz1 ~ R xyz0 + t``.
    The brush-layer level is bisected (deterministically) until the mean voxel count of the two
    clouds falls in ``band``; ``stats`` reports what was realised.
    """
    base = np.random.default_rng(seed)
    d = base.uniform(*dist_range)
    yaw = np.deg2rad(base.uniform(-max_yaw_deg, max_yaw_deg))
    lat = base.uniform(-0.5, 0.5)
    scene_seed, noise_seed = (int(v) for v in base.integers(0, 2**31, 2))
    poses = [_pose(0.0, 0.0, 0.0), _pose(d, lat, yaw)]
    scene = make_scene(np.random.default_rng(scene_seed))
    static = [_cast_static(scene, P, beams, azimuths, max_range, elev[0], elev[1],
                           np.random.default_rng([noise_seed, i])) for i, P in enumerate(poses)]
    lo, hi, level = 0.0, 6.0, 1.6
    for it in range(10):
        clouds = []
        for i, P in enumerate(poses):
            pts = _finish(P, static[i][0], static[i][1], level, np.random.default_rng([noise_seed, 7, i]),
                          max_range, 0.02)
            sel, c = voxelize(pts, voxel_size)
            clouds.append((pts, sel, c))
        mean_n = 0.5 * (len(clouds[0][1]) + len(clouds[1][1]))
        if band is None or band[0] <= mean_n <= band[1]:
            break
        if mean_n < band[0]:
            lo = level
        else:
            hi = level
        level = 0.5 * (lo + hi)
    (p0, sel0, c0), (p1, sel1, c1) = clouds
    T_gt = (np.linalg.inv(poses[1]) @ poses[0]).astype(np.float32)
    return {
        "xyz0": p0[sel0], "xyz1": p1[sel1],
        "coords0": c0, "coords1": c1,
        "feats0": np.ones((len(sel0), 1), np.float32), "feats1": np.ones((len(sel1), 1), np.float32),
        "T_gt": T_gt,
        "stats": {"raw0": len(p0), "raw1": len(p1), "n0": len(sel0), "n1": len(sel1),
                  "brush_level": level, "dist": float(d), "yaw_deg": float(np.rad2deg(yaw))},
    }


def batch_coords(coords_list):
    """``sparse_collate``-style batching: prepend the batch index, concatenate
    (``lib/data_loaders.py:65-66``)."""
    out = []
    for b, c in enumerate(coords_list):
        out.append(np.concatenate([np.full((len(c), 1), b, np.int32), c.astype(np.int32)], 1))
    return np.concatenate(out, 0)


def subsample_indices(seed, n, k=5000):
    """Seeded stand-in for ``np.random.choice(n, k, replace=False)`` of ``scripts/test_kitti.py:29-35``."""
    rng = np.random.default_rng(seed + 10**6)
    if n <= k:
        return np.arange(n)
    return rng.permutation(n)[:k]


def plant_correspondences(pair, seed, n_points=5000, inlier_ratio=0.3, plant_radius=0.3, feat_dim=32):
    """Sample sets and descriptors that give the matcher a STATED inlier ratio (benchmark "descriptor mode").

    The reference samples ``n_points`` voxels of each cloud uniformly (scripts/test_kitti.py:159-160) and a trained
    FCGF network then matches a fraction of them correctly.  Random-init weights carry no geometric signal, so the
    synthetic benchmark plants it: ``m = round(inlier_ratio * n_points)`` source voxels whose ground-truth image
    has a cloud-1 voxel within ``plant_radius`` metres are sampled together with that voxel (each target used
    once); the remaining ``n_points - m`` rows of both sample sets are uniform draws from the rest of the clouds.
    Planted partners share one random unit descriptor, every other row gets its own, so after the blend
    ``normalise(F_net + beta * G)`` the feature nearest neighbour of a planted source is its partner and every
    other source matches at random.  The residuals of the planted pairs are the real ones of the two voxelised
    scans (0 .. ``plant_radius``; the default is the RANSAC inlier threshold of 0.3 m, so "planted" = "inlier under
    the ground truth" and the residuals fill the whole inlier band, as a trained network's matches do - with a
    smaller radius every decent hypothesis catches ALL planted pairs and thousands of hypotheses tie on the count).

    Returns ``dict(sel0, sel1 int64 [n_points], G0, G1 f32 [n_points, feat_dim], planted int)``; ``planted`` can be
    below the request when the overlap is too small.
    """
    from scipy.spatial import cKDTree
    rng = np.random.default_rng([int(seed), 77])
    x0 = pair["xyz0"].astype(np.float64)
    x1 = pair["xyz1"].astype(np.float64)
    T = np.asarray(pair["T_gt"], np.float64)
    d, j = cKDTree(x1).query(x0 @ T[:3, :3].T + T[:3, 3], k=1)
    cand = np.nonzero(d < plant_radius)[0]
    cand = cand[rng.permutation(len(cand))]
    _, first = np.unique(j[cand], return_index=True)          # every target at most once
    cand = cand[np.sort(first)]
    n0, n1 = len(x0), len(x1)
    m = int(min(round(inlier_ratio * n_points), len(cand), n_points))
    src_pl, tgt_pl = cand[:m], j[cand[:m]]

    def fill(n, taken, count):
        rest = np.setdiff1d(np.arange(n), taken, assume_unique=False)
        if len(rest) >= count:
            return rest[rng.permutation(len(rest))[:count]]
        return rng.choice(n, count)                           # tiny clouds: with replacement, like random_sample
    sel0 = np.concatenate([src_pl, fill(n0, src_pl, n_points - m)])
    sel1 = np.concatenate([tgt_pl, fill(n1, tgt_pl, n_points - m)])

    def unit(k):
        g = rng.normal(size=(k, feat_dim))
        return (g / np.linalg.norm(g, axis=1, keepdims=True)).astype(np.float32)
    shared = unit(m)
    G0 = np.concatenate([shared, unit(n_points - m)])
    G1 = np.concatenate([shared, unit(n_points - m)])
    o0, o1 = rng.permutation(n_points), rng.permutation(n_points)   # planted rows anywhere in the sample sets
    return {"sel0": sel0[o0].astype(np.int64), "sel1": sel1[o1].astype(np.int64), "G0": G0[o0], "G1": G1[o1],
            "planted": m}


# (name, kernel volume, C_in, C_out) in the order ResUNet2.__init__ creates them
# (model/resunet.py:31-140 with the ResUNetBN2C tables at :206-209).
def RESUNET_BN2C_LAYOUT(in_channels=1, out_channels=32, conv1_kernel_size=5,
                        channels=(None, 32, 64, 128, 256), tr_channels=(None, 64, 64, 64, 128), expanded=False):
    C, T = channels, tr_channels
    k1 = conv1_kernel_size ** 3
    convs = [("conv1", k1, in_channels, C[1])]
    bns = [("norm1", C[1])]

    def block(name, c):
        convs.extend([(f"{name}.conv1", 27, c, c), (f"{name}.conv2", 27, c, c)])
        bns.extend([(f"{name}.norm1", c), (f"{name}.norm2", c)])
    block("block1", C[1])
    for i in (2, 3, 4):
        convs.append((f"conv{i}", 27, C[i - 1], C[i])); bns.append((f"norm{i}", C[i])); block(f"block{i}", C[i])
    convs.append(("conv4_tr", 27, C[4], T[4])); bns.append(("norm4_tr", T[4])); block("block4_tr", T[4])
    convs.append(("conv3_tr", 27, C[3] + T[4], T[3])); bns.append(("norm3_tr", T[3])); block("block3_tr", T[3])
    convs.append(("conv2_tr", 27, C[2] + T[3], T[2])); bns.append(("norm2_tr", T[2])); block("block2_tr", T[2])
    convs.append(("conv1_tr", 1, C[1] + T[2], T[1]))
    convs.append(("final", 1, T[1], out_channels))
    if expanded:       # ResUNetExpanded (model/resunet.py:254-425): norm<i>_2 + block<i>_2 per stage, appended so the plain layout's draws stay put
        last = convs.pop()
        for i, c in (("1", C[1]), ("2", C[2]), ("3", C[3]), ("4", C[4]), ("4_tr", T[4]), ("3_tr", T[3]), ("2_tr", T[2])):
            bns.append((f"norm{i}_2", c)); block(f"block{i}_2", c)
        convs.append(last)
    return convs, bns


def make_weights(seed=1234, **layout_kw):
    """Random ``state_dict`` with MinkowskiEngine's parameter names:
    kernels ~ N(0, 2/(K*C_in)), BN gamma ~ U[.5,1.5], beta, mean ~ N(0,.1), var ~ U[.5,1.5]."""
    rng = np.random.default_rng(seed)
    convs, bns = RESUNET_BN2C_LAYOUT(**layout_kw)
    sd = {}
    for name, K, ci, co in convs:
        std = np.sqrt(2.0 / (K * ci))
        w = rng.normal(0.0, std, (K, ci, co)).astype(np.float32)
        sd[f"{name}.kernel"] = w[0] if K == 1 else w
    sd["final.bias"] = rng.normal(0.0, 0.1, (1, convs[-1][3])).astype(np.float32)
    for name, c in bns:
        sd[f"{name}.bn.weight"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
        sd[f"{name}.bn.bias"] = rng.normal(0.0, 0.1, c).astype(np.float32)
        sd[f"{name}.bn.running_mean"] = rng.normal(0.0, 0.1, c).astype(np.float32)
        sd[f"{name}.bn.running_var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
        sd[f"{name}.bn.num_batches_tracked"] = np.array(0, np.int64)
    return sd
