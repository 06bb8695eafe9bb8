"""Input builders used by both ``make_golden.py`` and the tests (no reference import here)."""
import numpy as np

import _inputs as gi


def kabsch_inputs(seed, bs, n, kind):
    A = ((gi._u(seed, bs, n, 3) - 0.5) * 40.0)
    B = np.empty_like(A)
    for b in range(bs):
        tp = (gi._u(seed + 100 + b, 6) - 0.5) * np.array([0.6, 0.6, 2.0, 20, 20, 2])
        T = gi.rigid(*tp)
        B[b] = A[b] @ T[:3, :3].T + T[:3, 3]
    B = B + 0.05 * gi._normal(seed + 1, bs, n, 3)
    if kind == "reflect":
        B[:, :, 2] = -B[:, :, 2]            # best orthogonal fit is a reflection -> det fix path
    w = None
    if kind in ("weighted", "zeros", "reflect"):
        w = gi._u(seed + 2, bs, n)
        if kind == "zeros":
            w[gi._u(seed + 3, bs, n) < 0.5] = 0.0
    return A.astype(np.float32), B.astype(np.float32), None if w is None else w.astype(np.float32)
