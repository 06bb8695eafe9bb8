#!/usr/bin/env python3
"""CPU simulation (numpy, no GPU): how many (16-row chunk, offset) blocks of the staged stride-1 kernel are non-empty under
different ways of grouping a tile's rows into chunks, on the synthetic bench geometry (Z-ordered rows, 256-row tiles).

Reports, per level: mean occupied offsets per row / 27 (the floor), the non-empty block fraction of
  natural   rows in Z-order
  r4        the round-4 key (Gray rank of the 6 layer bits, then the 27-bit pattern, z layers first) = k_local_rulebook today
  greedy    seeded greedy clustering inside the tile (seed = densest unassigned row; add the row whose pattern adds fewest bits)
  pool<P>   the same, rows pooled over P consecutive Z-ordered rows before tiles are cut (distinct rows U per tile reported)
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
    """unique floor(c / ts) * ts in Z-order"""
    c = np.floor_divide(coords, ts)
    c = np.unique(c, axis=0)
    return c[np.argsort(morton(c), kind="stable")]


def nbr_table(c):
    """[27, N] row of the neighbour at offset k (x fastest) or -1; c is in units of the level's stride"""
    key = lambda q: ((q[:, 0] + (1 << 17)) << 36) | ((q[:, 1] + (1 << 17)) << 18) | (q[:, 2] + (1 << 17))
    k0 = key(c.astype(np.int64))
    order = np.argsort(k0)
    ks = k0[order]
    out = np.empty((27, len(c)), np.int64)
    i = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                q = key(c.astype(np.int64) + np.array([dx, dy, dz]))
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
    """number of non-empty (chunk, offset) blocks of rows given in chunk order (len multiple of 16 by padding with 0)"""
    n = len(mask_rows)
    pad = (-n) % 16
    m = np.concatenate([mask_rows, np.zeros(pad, np.int64)]).reshape(-1, 16)
    u = np.bitwise_or.reduce(m, axis=1)
    return int(popc(u).sum()), len(u)


def key_r4(mask):
    zp, zm, z0 = (mask >> 18) & 0x1FF, mask & 0x1FF, (mask >> 9) & 0x1FF
    yp = (mask & 0x70381C0) != 0
    ym = (mask & 0x01C0E07) != 0
    xp = (mask & 0x4924924) != 0
    xm = (mask & 0x1249249) != 0
    cb = (zp != 0) * 32 | (zm != 0) * 16 | yp * 8 | ym * 4 | xp * 2 | xm * 1
    cb = cb ^ (cb >> 1)
    cb = cb ^ (cb >> 2)
    cb = cb ^ (cb >> 4)
    pattern = (zp << 18) | (zm << 9) | z0
    return (cb.astype(np.int64) << 35) | (pattern << 8)


def greedy(mask, seed_mode="dense"):
    """order of rows: chunks built one after the other; returns permutation"""
    n = len(mask)
    left = np.ones(n, bool)
    pc = popc(mask)
    order = []
    while left.any():
        idx = np.flatnonzero(left)
        if seed_mode == "dense":
            s = idx[np.argmax(pc[idx])]
        else:
            s = idx[np.argmin(pc[idx])]
        u = mask[s]
        left[s] = False
        order.append(s)
        for _ in range(15):
            idx = np.flatnonzero(left)
            if len(idx) == 0:
                break
            add = popc(mask[idx] & ~u)
            # fewest added bits, then most bits shared with the union
            cost = add * 64 - popc(mask[idx] & u)
            b = idx[np.argmin(cost)]
            u |= mask[b]
            left[b] = False
            order.append(b)
    return np.array(order)


def distinct(nbr, rows):
    v = nbr[:, rows]
    return len(np.unique(v[v >= 0]))


def main():
    seeds = [int(s) for s in sys.argv[1:]] or [3]
    for seed in seeds:
        pair = synthetic.make_pair(seed)
        c0 = pair["coords0"].astype(np.int64)
        for lvl in range(4):
            c = level_rows(c0, 1 << lvl)
            nbr = nbr_table(c)
            mask = masks_of(nbr)
            n = len(c)
            res = {}
            tot_nat = tot_r4 = tot_gr = tot_ch = 0
            for t0 in range(0, n, 256):
                m = mask[t0:t0 + 256]
                b, ch = blocks(m); tot_nat += b; tot_ch += ch
                o = np.argsort(key_r4(m) | np.arange(len(m)), kind="stable")
                tot_r4 += blocks(m[o])[0]
                tot_gr += blocks(m[greedy(m)])[0]
            res["natural"] = tot_nat / (27 * tot_ch)
            res["r4"] = tot_r4 / (27 * tot_ch)
            res["greedy"] = tot_gr / (27 * tot_ch)
            line = f"seed {seed} level {lvl}: rows {n:6d}  floor {popc(mask).mean() / 27:.3f}  " + "  ".join(f"{k} {v:.3f}" for k, v in res.items())
            for P in (512, 1024):
                tot = 0; us = []
                for p0 in range(0, n, P):
                    m = mask[p0:p0 + P]
                    o = greedy(m)
                    tot += blocks(m[o])[0]
                    for t0 in range(0, len(o), 256):
                        us.append(distinct(nbr, p0 + o[t0:t0 + 256]))
                line += f"  pool{P} {tot / (27 * tot_ch):.3f} (U mean {np.mean(us):.0f} max {np.max(us)})"
            us = [distinct(nbr, np.arange(t0, min(t0 + 256, n))) for t0 in range(0, n, 256)]
            line += f"  [U tile mean {np.mean(us):.0f} max {np.max(us)}]"
            print(line, flush=True)


if __name__ == "__main__":
    main()
