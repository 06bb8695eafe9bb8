"""CPU oracle of the voxeliser (numpy).  TEST INFRASTRUCTURE.

**Parity unpinned** for the choice of representative point: ``ME.utils.sparse_quantize`` is part of
MinkowskiEngine (absent here).  Restated semantics (lib/data_loaders.py:940-943,969-979;
util/misc.py:80-84): voxel = ``floor(xyz / voxel_size)`` evaluated in fp32, one point per occupied
voxel - the first in input order - with the selection returned in ascending order."""
import numpy as np


def sparse_quantize(xyz, voxel_size, batch_index=0):
    c = np.floor(np.asarray(xyz, np.float32)[:, :3] / np.float32(voxel_size)).astype(np.int64)
    _, first = np.unique(c, axis=0, return_index=True)
    sel = np.sort(first)
    coords = np.concatenate([np.full((len(sel), 1), batch_index, np.int64), c[sel]], 1).astype(np.int32)
    return coords, sel.astype(np.int64)
