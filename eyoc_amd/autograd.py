"""Differentiable sparse convolution and the hardest-contrastive loss of EYOC's trainer.

``sparse_conv`` is one ``MinkowskiConvolution`` / ``MinkowskiConvolutionTranspose`` (no bias, like every convolution of
``ResUNet2`` except ``final``: model/resunet.py:31-140) as a ``torch.autograd.Function`` whose forward AND backward run
in ``libeyoc_hip.so``: grad-input is the forward kernel over the transposed rulebook with transposed weights, grad-weight
is ``eyoc_spconv_grad_weight`` (fp32 MFMA over the gathered pair lists).  ``contrastive_hardest_negative_loss`` mirrors
``lib/trainer.py:935-991``: the hardest negatives come from ``eyoc_knn1`` (nothing N x M is materialised), the loss
itself is a handful of torch element-wise ops on the gathered rows, so ``loss.backward()`` (lib/trainer.py:1667) works.

Training-mode batch norm, the optimiser and the trainer loop stay out of scope."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .eval import knn1_segmented


def _pack(weight: torch.Tensor, transposed: bool, mirror: bool) -> torch.Tensor:
    lib = _lib.load()
    w = np.ascontiguousarray(weight.detach().cpu().numpy().astype(np.float32))
    K, cin, cout = w.shape
    packed = np.zeros(w.size, np.float32)
    if transposed:
        rc = lib.eyoc_spconv_pack_weights_transposed(w.ctypes.data, K, cin, cout, 1 if mirror else 0, packed.ctypes.data)
    else:
        rc = lib.eyoc_spconv_pack_weights(w.ctypes.data, None, K, cin, cout, packed.ctypes.data)
    _lib.check(rc, "eyoc_spconv_pack_weights")
    return torch.from_numpy(packed).to(weight.device)


def _run(table, n_out, x, packed, cin, cout):
    lib = _lib.load()
    out = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
    K = 1 if table is None else table.shape[0]
    with torch.cuda.device(x.device):
        _lib.check(lib.eyoc_spconv(_lib.ctx(x.device.index), _lib.ptr(table), K, n_out, _lib.ptr(x), x.stride(0), cin, _lib.ptr(packed),
                                   cout, None, None, 0, 0, _lib.ptr(out), out.stride(0), _lib.stream_ptr()), "eyoc_spconv")
    return out


class _SparseConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, table, table_t, mirror, n_out):
        x = x.contiguous()
        K, cin, cout = weight.shape
        ctx.save_for_backward(x, weight)
        ctx.table, ctx.table_t, ctx.mirror, ctx.n_in = table, table_t, mirror, x.shape[0]
        return _run(table, n_out, x, _pack(weight, False, False), cin, cout)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        K, cin, cout = weight.shape
        lib = _lib.load()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            # the same operator over the transposed rulebook with W[k]^T (mirrored offsets for a self-transposed table)
            dx = _run(ctx.table_t, ctx.n_in, dy, _pack(weight, True, ctx.mirror), cout, cin)
        if ctx.needs_input_grad[1]:
            n_out = dy.shape[0]
            dw = torch.empty_like(weight, dtype=torch.float32)
            with torch.cuda.device(x.device):
                ws = _lib.workspace(lib.eyoc_spconv_grad_weight_workspace_bytes(K, n_out, cin, cout), x.device)
                _lib.check(lib.eyoc_spconv_grad_weight(_lib.ctx(x.device.index), _lib.ptr(ctx.table), K, n_out, _lib.ptr(x), x.stride(0),
                                                       cin, _lib.ptr(dy), dy.stride(0), cout, _lib.ptr(dw), _lib.ptr(ws), ws.numel(),
                                                       _lib.stream_ptr()), "eyoc_spconv_grad_weight")
        return dx, dw, None, None, None, None


def sparse_conv(x: torch.Tensor, weight: torch.Tensor, table: torch.Tensor | None, table_t: torch.Tensor | None = None,
                n_out: int | None = None) -> torch.Tensor:
    """``out[o] = sum_k x[table[k][o]] @ weight[k]`` with autograd.

    ``table`` int32 ``[K, n_out]`` on the GPU (``CoordinateManager.table(kind, level)``), ``None`` = identity (1x1).
    ``table_t``: the transposed rulebook for the gradient w.r.t. ``x`` - omit it for a stride-1 table (it is its own
    transpose under mirrored offsets); for the strided table of a level pass that level's transposed (up) table and
    vice versa.  Channel counts follow ``eyoc_spconv`` (C_in % 32 == 0, C_out in {32, 64, 128, 256}); for the gradient
    w.r.t. ``x`` the same must hold with the roles swapped."""
    mirror = table_t is None
    if table is not None and n_out is None:
        n_out = table.shape[1]
    if table is None:
        n_out = x.shape[0]
    return _SparseConv.apply(x, weight, table, table if mirror else table_t, mirror, n_out)


def contrastive_hardest_negative_loss(F0, F1, positive_pairs, num_pos=5192, num_hn_samples=2048, pos_thresh=0.1, neg_thresh=1.4,
                                      rng=None):
    """lib/trainer.py:935-991 -> ``(pos_loss, neg_loss)`` (0-dim tensors with grad).

    ``rng`` replaces the reference's global ``np.random`` (three ``choice`` draws: the two hard-negative candidate
    sets, then the positive subsample).  The nearest candidate of every positive comes from ``eyoc_knn1`` in "L2" mode
    (``sqrt(d2 + 1e-7)``, ties to the lowest index); its distance is re-evaluated with torch ops on the two gathered
    rows so that the gradient reaches both ends, exactly where ``D01.min(1)`` sends it."""
    rng = np.random if rng is None else rng
    N0, N1 = len(F0), len(F1)
    pp = positive_pairs if isinstance(positive_pairs, torch.Tensor) else torch.as_tensor(np.asarray(positive_pairs))
    N_pos_pairs = len(pp)
    hash_seed = max(N0, N1)
    sel0 = rng.choice(N0, min(N0, num_hn_samples), replace=False)
    sel1 = rng.choice(N1, min(N1, num_hn_samples), replace=False)
    if N_pos_pairs > num_pos:
        pos_sel = rng.choice(N_pos_pairs, num_pos, replace=False)
        sample_pos_pairs = pp[torch.as_tensor(pos_sel)]
    else:
        sample_pos_pairs = pp
    dev = F0.device
    sel0_d, sel1_d = torch.as_tensor(sel0).to(dev), torch.as_tensor(sel1).to(dev)
    pos_ind0 = sample_pos_pairs[:, 0].long().to(dev)
    pos_ind1 = sample_pos_pairs[:, 1].long().to(dev)
    posF0, posF1 = F0[pos_ind0], F1[pos_ind1]
    subF0, subF1 = F0[sel0_d], F1[sel1_d]
    with torch.no_grad():
        n_pos, c = posF0.shape
        i01 = knn1_segmented(posF0.detach(), subF1.detach(), [0, n_pos], [0, len(sel1)], "L2", return_distance=False)
        i10 = knn1_segmented(posF1.detach(), subF0.detach(), [0, n_pos], [0, len(sel0)], "L2", return_distance=False)
    D01min = torch.sqrt((posF0 - subF1[i01]).pow(2).sum(1) + 1e-7)
    D10min = torch.sqrt((posF1 - subF0[i10]).pow(2).sum(1) + 1e-7)
    # hardest negatives that are themselves positives are masked out (hash of the index pair, like util/misc.py:6-18)
    pos_keys = pp[:, 0].long().cpu().numpy() + pp[:, 1].long().cpu().numpy() * hash_seed
    neg0 = pos_ind0.cpu().numpy() + sel1[i01.cpu().numpy()] * hash_seed
    neg1 = sel0[i10.cpu().numpy()] + pos_ind1.cpu().numpy() * hash_seed
    mask0 = torch.from_numpy(np.logical_not(np.isin(neg0, pos_keys))).to(dev)
    mask1 = torch.from_numpy(np.logical_not(np.isin(neg1, pos_keys))).to(dev)
    pos_loss = torch.relu((posF0 - posF1).pow(2).sum(1) - pos_thresh)
    neg_loss0 = torch.relu(neg_thresh - D01min[mask0]).pow(2)
    neg_loss1 = torch.relu(neg_thresh - D10min[mask1]).pow(2)
    return pos_loss.mean(), (neg_loss0.mean() + neg_loss1.mean()) / 2
