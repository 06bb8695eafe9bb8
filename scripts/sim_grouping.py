#!/usr/bin/env python3
"""CPU simulation (numpy, no GPU): where is the floor of the staged kernels' zero-row products?

The staged stride-1 kernel (csrc/spconv_st.hip) multiplies a (16-row chunk, offset) block whenever ANY row of the chunk has a
neighbour at the offset; the builder (k_local_rulebook) sorts a tile's 256 rows by a key of their neighbour pattern so that a chunk's
rows miss the same offsets.  This script rebuilds the stride-1 tables of synthetic bench clouds in Z-order and reports, per level, the
fraction of non-empty blocks under
  floor     mean occupied offsets per row / 27 (what a kernel with 1-row granularity would multiply)
  natural   rows in Z-order
  key       the builder's key (Gray rank of the 6 layer bits, then the 27-bit pattern, z layers first) - what runs
  search    a local search (30 000 accepted-or-rejected row swaps per tile) started from `key` - how far a better grouping INSIDE a tile
            could go
  pool P    `key` applied to P consecutive rows before tiles are cut (P = 256 is `key`), with the mean number of distinct input rows U a
            256-row tile then stages (the kernel's stage holds 639 per pass, 1278 in two) - how far a wider grouping pool could go
and, for the strided tables (round 6: staged on 128-row output tiles), the distinct fine rows per 128- / 256-row coarse tile and the
non-empty fractions with the second stage pass counted.

usage: python scripts/sim_grouping.py [seed ...] > profiles/r6_grouping_floor.txt
"""
import sys
import numpy as np

sys.path.insert(0, ".")
from eyoc_amd import synthetic  # noqa: E402


def morton(c):
    c = c.astype(np.int64) + (1 << 17)
    def spread(v):
        out = np.zeros_like(v)
        for b in range(18):
            out |= ((v >> b) & 1) << (3 * b)
        return out
    return spread(c[:, 0]) | (spread(c[:, 1]) << 1) | (spread(c[:, 2]) << 2)


def level_rows(coords, ts):
    """unique floor(c / ts) in Z-order (units of the level's stride)"""
    c = np.unique(np.floor_divide(coords, ts), axis=0)
    return c[np.argsort(morton(c), kind="stable")]


def _key(q):
    return ((q[:, 0] + (1 << 17)) << 36) | ((q[:, 1] + (1 << 17)) << 18) | (q[:, 2] + (1 << 17))


def table(c_in, c_out, scale):
    """[27, n_out]: row of c_in at scale * c_out + off (x fastest) or -1"""
    k0 = _key(c_in.astype(np.int64))
    order = np.argsort(k0)
    ks = k0[order]
    out = np.empty((27, len(c_out)), np.int64)
    i = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                q = _key(scale * c_out.astype(np.int64) + np.array([dx, dy, dz]))
                pos = np.minimum(np.searchsorted(ks, q), len(ks) - 1)
                out[i] = np.where(ks[pos] == q, order[pos], -1)
                i += 1
    return out


def masks_of(nbr):
    m = np.zeros(nbr.shape[1], np.int64)
    for k in range(27):
        m |= (nbr[k] >= 0).astype(np.int64) << k
    return m


def popc(x):
    x = np.asarray(x, np.int64)
    n = np.zeros(x.shape, np.int64)
    for b in range(27):
        n += (x >> b) & 1
    return n


def blocks(mask_rows):
    """(non-empty (chunk, offset) blocks, chunks) of rows given in chunk order"""
    pad = (-len(mask_rows)) % 16
    m = np.concatenate([mask_rows, np.zeros(pad, np.int64)]).reshape(-1, 16)
    return int(popc(np.bitwise_or.reduce(m, axis=1)).sum()), len(m)


def key_r4(mask):
    zp, zm, z0 = (mask >> 18) & 0x1FF, mask & 0x1FF, (mask >> 9) & 0x1FF
    yp, ym = (mask & 0x70381C0) != 0, (mask & 0x01C0E07) != 0
    xp, xm = (mask & 0x4924924) != 0, (mask & 0x1249249) != 0
    cb = (zp != 0) * 32 | (zm != 0) * 16 | yp * 8 | ym * 4 | xp * 2 | xm * 1
    cb = cb ^ (cb >> 1)
    cb = cb ^ (cb >> 2)
    cb = cb ^ (cb >> 4)
    return (cb.astype(np.int64) << 35) | (((zp << 18) | (zm << 9) | z0) << 8)


def by_key(m):
    return np.argsort(key_r4(m) | np.arange(len(m)), kind="stable")


def local_search(m, order, rng, iters=30000):
    nch = (len(order) + 15) // 16
    mm = np.concatenate([m[order], np.zeros(nch * 16 - len(order), np.int64)]).reshape(nch, 16)
    cost = lambda ch: int(popc(np.bitwise_or.reduce(ch)))
    costs = [cost(mm[i]) for i in range(nch)]
    for _ in range(iters):
        a, b = rng.integers(0, nch, 2)
        if a == b:
            continue
        i, j = rng.integers(0, 16, 2)
        mm[a, i], mm[b, j] = mm[b, j], mm[a, i]
        ca, cb = cost(mm[a]), cost(mm[b])
        if ca + cb <= costs[a] + costs[b]:
            costs[a], costs[b] = ca, cb
        else:
            mm[a, i], mm[b, j] = mm[b, j], mm[a, i]
    return sum(costs)


def distinct(nbr, rows):
    v = nbr[:, rows]
    return len(np.unique(v[v >= 0]))


def main():
    seeds = [int(s) for s in sys.argv[1:]] or [3, 5]
    rng = np.random.default_rng(0)
    print(__doc__.split("usage:")[0].strip().replace("\n", "\n# ").replace("\"\"\"", ""))
    print()
    for seed in seeds:
        c0 = synthetic.make_pair(seed)["coords0"].astype(np.int64)
        lv = [level_rows(c0, 1 << l) for l in range(4)]
        print(f"== cloud of seed {seed}: rows per level {[len(c) for c in lv]}")
        print("stride-1 tables (256-row tiles)")
        for l in range(4):
            nbr = table(lv[l], lv[l], 1)
            mask = masks_of(nbr)
            n = len(mask)
            tot = {"natural": 0, "key": 0, "search": 0}
            ch = ch_s = 0
            for ti, t0 in enumerate(range(0, n, 256)):
                m = mask[t0:t0 + 256]
                b, c_ = blocks(m); tot["natural"] += b; ch += c_
                o = by_key(m)
                bk = blocks(m[o])[0]; tot["key"] += bk
                if ti % 6 == 0:                                  # the search on every sixth tile
                    tot["search"] += local_search(m, o, rng) - bk * 0; ch_s += c_
                    tot.setdefault("key_s", 0); tot["key_s"] += bk
            line = (f"  level {l}: rows {n:6d}  floor {popc(mask).mean() / 27:.3f}  natural {tot['natural'] / (27 * ch):.3f}  key {tot['key'] / (27 * ch):.3f}"
                    f"  search {tot['search'] / (27 * ch_s):.3f} (key on the same tiles {tot['key_s'] / (27 * ch_s):.3f})")
            for P in (512, 1024, 4096, 1 << 30):
                t = c_all = 0
                us = []
                for p0 in range(0, n, P):
                    m = mask[p0:p0 + P]
                    o = by_key(m)
                    b, c_ = blocks(m[o]); t += b; c_all += c_
                    if P <= 4096:
                        us += [distinct(nbr, p0 + o[t0:t0 + 256]) for t0 in range(0, len(o), 256)]
                line += f"  pool {P if P < 1 << 30 else 'all'} {t / (27 * c_all):.3f}" + (f" (U {np.mean(us):.0f})" if us else "")
            u0 = [distinct(nbr, np.arange(t0, min(t0 + 256, n))) for t0 in range(0, n, 256)]
            print(line + f"  [U of a Z-order tile: mean {np.mean(u0):.0f}, max {np.max(u0)}]", flush=True)
        print("strided tables (outputs = the coarser level's rows)")
        for l in range(3):
            nbr = table(lv[l], lv[l + 1], 2)
            mask = masks_of(nbr)
            n = len(mask)
            line = f"  {l} -> {l + 1}: coarse rows {n:6d}  pairs per row {popc(mask).mean():.2f}"
            for T in (128, 256):
                us = []
                tot = tot2 = ch = 0
                for t0 in range(0, n, T):
                    rows = np.arange(t0, min(t0 + T, n))
                    m = mask[rows]
                    o = by_key(m)
                    b, c_ = blocks(m[o]); tot += b; ch += c_
                    v = nbr[:, rows]
                    u = np.unique(v[v >= 0]); us.append(len(u))
                    if len(u) > 639:                             # second stage pass: the rows past the 639th are staged later; blocks per pass
                        thr = u[639]
                        for sel in (lambda x: (x >= 0) & (x < thr), lambda x: x >= thr):
                            mm = np.zeros(len(rows), np.int64)
                            for k in range(27):
                                mm |= sel(v[k]).astype(np.int64) << k
                            tot2 += blocks(mm[o])[0]
                    else:
                        tot2 += b
                us = np.array(us)
                line += (f"  | {T}-row tiles: U mean {us.mean():.0f} p90 {np.percentile(us, 90):.0f} max {us.max()}, two passes {(us > 639).mean():.2f},"
                         f" overflow (> 1278) {(us > 1278).mean():.2f}, non-empty {tot / (27 * ch):.3f} -> {tot2 / (27 * ch):.3f} with the second pass")
            print(line, flush=True)


if __name__ == "__main__":
    main()
