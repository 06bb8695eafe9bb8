"""Coordinate algebra of a MinkowskiEngine-style sparse tensor (CPU oracle, numpy).

TEST INFRASTRUCTURE.  **Parity unpinned**: MinkowskiEngine (README.md:27,61 of the reference - "v0.5
or higher", installed from git HEAD, not vendored) is absent, so these functions restate its
published semantics and are pinned by dense-convolution equivalence tests only.

Call sites being restated:
  * ``ME.SparseTensor(features, coordinates=)``            scripts/test_kitti.py:143-148
  * ``ME.MinkowskiConvolution(kernel_size=3|5, stride=1)``  model/resunet.py:31-38, model/residual_block.py:23-33
  * ``ME.MinkowskiConvolution(kernel_size=3, stride=2)``    model/resunet.py:44-77
  * ``ME.MinkowskiConvolutionTranspose(kernel_size=3, stride=2)``  model/resunet.py:83-116

Semantics:
  (1) stride-1 conv: output coordinates == input coordinates; input row i feeds output row o
      through kernel offset k iff ``c_in[i] == c_out[o] + off_k * ts``.
  (2) stride-2 conv from tensor stride ts: output coordinates = unique ``floor(c / 2ts) * 2ts``;
      offsets ``off_k * ts`` around the OUTPUT coordinate.
  (3) transposed stride-2 conv from 2ts to ts: outputs live on the existing ts map; its pairs are
      the transpose of the forward (ts -> 2ts) map with the same kernel index.
  (4) offsets enumerate x fastest: ``k = (dx+r) + K*(dy+r) + K*K*(dz+r)``.
  (5) row order of a derived map = order of first occurrence.

A map is represented as ``nbr[K, N_out] int32`` holding the input row for each (offset, output row)
or -1.  This "output-stationary" table is exactly what the HIP rulebook builder emits.
"""
from __future__ import annotations

import numpy as np

def kernel_offsets(kernel_size: int) -> np.ndarray:
    """``[K^3, 3]`` integer offsets, x fastest (item 4 above)."""
    r = kernel_size // 2
    rng = range(-r, r + 1)
    return np.array([(dx, dy, dz) for dz in rng for dy in rng for dx in rng], dtype=np.int64)


class CoordMap:
    """Unique coordinate set at one tensor stride, with row lookup."""

    def __init__(self, coords: np.ndarray, tensor_stride: int = 1):
        self.coords = np.ascontiguousarray(coords, dtype=np.int64)
        self.ts = int(tensor_stride)
        keys = self._key(self.coords)
        self._order = np.argsort(keys, kind="stable")
        self._sorted = keys[self._order]
        if len(keys) > 1 and np.any(self._sorted[1:] == self._sorted[:-1]):
            raise ValueError("duplicate coordinates in sparse tensor")

    @staticmethod
    def _key(c):
        c = c.astype(np.int64)
        return (c[:, 0] << 54) | (((c[:, 1] + (1 << 17)) & 0x3FFFF) << 36) | \
               (((c[:, 2] + (1 << 17)) & 0x3FFFF) << 18) | ((c[:, 3] + (1 << 17)) & 0x3FFFF)

    def __len__(self):
        return len(self.coords)

    def lookup(self, q: np.ndarray) -> np.ndarray:
        """Row index of each query coordinate ``[M,4]`` or -1."""
        if len(self.coords) == 0:
            return np.full(len(q), -1, np.int32)
        k = self._key(q)
        pos = np.searchsorted(self._sorted, k)
        pos = np.minimum(pos, len(self._sorted) - 1)
        hit = self._sorted[pos] == k
        return np.where(hit, self._order[pos], -1).astype(np.int32)


def stride_map(cm: CoordMap, stride: int = 2) -> tuple[CoordMap, np.ndarray]:
    """Down-sampled coordinate map (item 2) and ``parent[N_in]`` = coarse row of every fine row."""
    nts = cm.ts * stride
    c = cm.coords.copy()
    c[:, 1:] = np.floor_divide(c[:, 1:], nts) * nts
    keys = CoordMap._key(c)
    _, first, inv = np.unique(keys, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")          # coarse rows in order of first occurrence
    rank = np.empty_like(order)
    rank[order] = np.arange(len(order))
    return CoordMap(c[first[order]], nts), rank[inv].astype(np.int32)


def kernel_map(cm_in: CoordMap, cm_out: CoordMap, kernel_size: int) -> np.ndarray:
    """``nbr[K, N_out]``: input row at ``c_out + off_k * ts_in`` (items 1 and 2)."""
    offs = kernel_offsets(kernel_size) * cm_in.ts
    nbr = np.empty((len(offs), len(cm_out)), np.int32)
    q = cm_out.coords.copy()
    for k, off in enumerate(offs):
        q[:, 1:] = cm_out.coords[:, 1:] + off[None, :]
        nbr[k] = cm_in.lookup(q)
    return nbr


def transposed_kernel_map(cm_coarse: CoordMap, cm_fine: CoordMap, kernel_size: int) -> np.ndarray:
    """``nbr[K, N_fine]`` for the transposed conv (item 3): the forward pair (u -> v, k) exists iff
    ``c_u == c_v + off_k * ts_fine``; the transposed conv accumulates ``in[v] W[k]`` into ``out[u]``,
    so for fine row u and offset k the source is the coarse row at ``c_u - off_k * ts_fine``."""
    offs = kernel_offsets(kernel_size) * cm_fine.ts
    nbr = np.empty((len(offs), len(cm_fine)), np.int32)
    q = cm_fine.coords.copy()
    for k, off in enumerate(offs):
        q[:, 1:] = cm_fine.coords[:, 1:] - off[None, :]
        nbr[k] = cm_coarse.lookup(q)
    return nbr


def build_maps(coords: np.ndarray, conv1_kernel_size: int = 5, levels: int = 4) -> dict:
    """Everything ``ResUNet2.forward`` needs for one (batched) coordinate set ``[N,4] int``.

    Returns ``{"cm": [cm_1, cm_2, cm_4, cm_8], "k5": nbr, "s1": [nbr per level],
    "down": [1->2, 2->4, 4->8], "up": [2->1, 4->2, 8->4]}``.
    """
    cms = [CoordMap(np.asarray(coords), 1)]
    for _ in range(levels - 1):
        cms.append(stride_map(cms[-1], 2)[0])
    out = {"cm": cms}
    out["k5"] = kernel_map(cms[0], cms[0], conv1_kernel_size)
    out["s1"] = [kernel_map(c, c, 3) for c in cms]
    out["down"] = [kernel_map(cms[i], cms[i + 1], 3) for i in range(levels - 1)]
    out["up"] = [transposed_kernel_map(cms[i + 1], cms[i], 3) for i in range(levels - 1)]
    return out


def map_stats(maps: dict) -> dict:
    """Realised sizes / pair counts (feeds the algorithmic work model of SURVEY.md §8d)."""
    n = [len(c) for c in maps["cm"]]
    pairs = lambda t: int((t >= 0).sum())
    return {
        "rows": n,
        "pairs_k5": pairs(maps["k5"]),
        "pairs_s1": [pairs(t) for t in maps["s1"]],
        "pairs_down": [pairs(t) for t in maps["down"]],
        "pairs_up": [pairs(t) for t in maps["up"]],
    }
