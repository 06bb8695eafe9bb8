"""CPU oracle of the hardest-contrastive loss and of the sparse-convolution gradients.  TEST INFRASTRUCTURE.

* ``contrastive_hardest_negative_loss`` restates ``lib/trainer.py:935-991`` with torch CPU ops (dense ``pdist`` of
  ``lib/metrics.py:22-29``, ``min(1)``, the index-pair hash of ``util/misc.py:6-18``, ``np.isin`` masks, relu losses);
  the three ``np.random.choice`` draws come from the ``rng`` argument in the reference's order.  **Parity unpinned**
  (``lib/trainer.py`` cannot be imported: MinkowskiEngine / pytorch3d / open3d at module scope) except for ``pdist``
  and ``_hash``, which the CPU tests check against the reference's own functions when its tree is present.
* gradients of a sparse convolution: autograd through ``oracle.resunet.sparse_conv`` (gather -> matmul -> index_add).
"""
from __future__ import annotations

import numpy as np
import torch


def pdist(A, B, dist_type="L2"):
    D2 = torch.sum((A.unsqueeze(1) - B.unsqueeze(0)).pow(2), 2)
    return torch.sqrt(D2 + 1e-7) if dist_type == "L2" else D2


def pair_hash(arr, M):
    """util/misc.py:6-18: ``sum_d arr[d] * M**d`` over the columns (int64)."""
    cols = [arr[:, d] for d in range(arr.shape[1])] if isinstance(arr, np.ndarray) else list(arr)
    h = np.zeros(len(cols[0]), np.int64)
    for d, c in enumerate(cols):
        h += np.asarray(c, np.int64) * M ** d
    return h


def contrastive_hardest_negative_loss(F0, F1, positive_pairs, num_pos=5192, num_hn_samples=2048, pos_thresh=0.1,
                                      neg_thresh=1.4, rng=None):
    rng = np.random if rng is None else rng
    N0, N1 = len(F0), len(F1)
    positive_pairs = np.asarray(positive_pairs, np.int64)
    hash_seed = max(N0, N1)
    sel0 = rng.choice(N0, min(N0, num_hn_samples), replace=False)
    sel1 = rng.choice(N1, min(N1, num_hn_samples), replace=False)
    if len(positive_pairs) > num_pos:
        sample = positive_pairs[rng.choice(len(positive_pairs), num_pos, replace=False)]
    else:
        sample = positive_pairs
    subF0, subF1 = F0[sel0], F1[sel1]
    pos_ind0, pos_ind1 = torch.from_numpy(sample[:, 0]), torch.from_numpy(sample[:, 1])
    posF0, posF1 = F0[pos_ind0], F1[pos_ind1]
    D01min, D01ind = pdist(posF0, subF1).min(1)
    D10min, D10ind = pdist(posF1, subF0).min(1)
    pos_keys = pair_hash(positive_pairs, hash_seed)
    neg_keys0 = pair_hash([pos_ind0.numpy(), sel1[D01ind.numpy()]], hash_seed)
    neg_keys1 = pair_hash([sel0[D10ind.numpy()], pos_ind1.numpy()], hash_seed)
    mask0 = torch.from_numpy(np.logical_not(np.isin(neg_keys0, pos_keys)))
    mask1 = torch.from_numpy(np.logical_not(np.isin(neg_keys1, pos_keys)))
    pos_loss = torch.relu((posF0 - posF1).pow(2).sum(1) - pos_thresh)
    neg_loss0 = torch.relu(neg_thresh - D01min[mask0]).pow(2)
    neg_loss1 = torch.relu(neg_thresh - D10min[mask1]).pow(2)
    return pos_loss.mean(), (neg_loss0.mean() + neg_loss1.mean()) / 2
