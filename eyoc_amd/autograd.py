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


_PERM_CACHE: dict = {}      # (K, cin, cout, transposed, mirror, device) -> int64 gather indices; one entry per layer SHAPE, never stale


def _pack_perm(K, cin, cout, transposed, mirror, device):
    """The library's host packers only MOVE values (fragment order of the MFMA kernels, zeros in padding slots): packing
    the sequence 1, 2, 3 ... once per layer shape yields the permutation, and every later pack is one gather on the device."""
    key = (K, cin, cout, bool(transposed), bool(mirror), str(device))
    if key not in _PERM_CACHE:
        lib = _lib.load()
        n = K * cin * cout

        def run(w):
            packed = np.zeros(n, np.float32)
            if transposed:
                rc = lib.eyoc_spconv_pack_weights_transposed(w.ctypes.data, K, cin, cout, 1 if mirror else 0, packed.ctypes.data)
            else:
                rc = lib.eyoc_spconv_pack_weights(w.ctypes.data, None, K, cin, cout, packed.ctypes.data)
            _lib.check(rc, "eyoc_spconv_pack_weights")
            return packed.astype(np.int64)

        # fp32 holds integers exactly up to 2^24: the element number travels as two sequences (i // 4096 + 1, 0 = padding; i % 4096)
        idx = np.arange(n, dtype=np.int64)
        hi, lo = run((idx // 4096 + 1).astype(np.float32)), run((idx % 4096).astype(np.float32))
        packed = np.where(hi > 0, (hi - 1) * 4096 + lo + 1, 0)
        # 0 = padding slot, i + 1 = element i.  Layouts without padding slots (every shape of the ResUNet tables) are kept as plain
        # indices: the pack is then ONE gather instead of zeros + cat + gather
        if (packed > 0).all():
            _PERM_CACHE[key] = (torch.from_numpy(packed - 1).to(device), False)
        else:
            _PERM_CACHE[key] = (torch.from_numpy(packed).to(device), True)
    return _PERM_CACHE[key]


def _pack(weight: torch.Tensor, transposed: bool, mirror: bool) -> torch.Tensor:
    """Weights in the MFMA fragment order the kernels consume: a device-side gather through the shape's permutation - no
    host round trip, no synchronisation, nothing cached that depends on the parameter's contents."""
    K, cin, cout = weight.shape
    perm, padded = _pack_perm(K, cin, cout, transposed, mirror, weight.device)
    with torch.no_grad():
        flat = weight.detach().reshape(-1).float()
        if not padded:
            return flat[perm]
        return torch.cat([flat.new_zeros(1), flat])[perm]


def _run(table, n_out, x, packed, cin, cout, out=None):
    """``out``: a ``[n_out, cout]`` view to write into (rows ``out.stride(0)`` floats apart: a column block of a wider tensor)."""
    lib = _lib.load()
    if out is None:
        out = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
    K = 1 if table is None else table.shape[0]
    with _lib.on_device(x.device):
        _lib.check(lib.eyoc_spconv_sum(_lib.ctx(x.device.index), _lib.ptr(table), K, n_out, _lib.ptr(x), x.stride(0), cin, _lib.ptr(packed),
                                       cout, _lib.ptr(out), out.stride(0), _lib.stream_ptr()), "eyoc_spconv_sum")
    return out


_BULK_CACHE: dict = {}


def bulk_pack(kernels):
    """The packed forms of many layer kernels at once: ``kernels`` = [(weight [K, C_in, C_out], mirror)] -> [(packed, packed_transposed or
    None)] through ONE concatenation and ONE gather (a training iteration packs 21 kernels twice: 42 gathers of ~15 us host time each
    otherwise).  ``packed_transposed`` is ``None`` where the input gradient runs in column blocks (C_in not 32 / 64 / 128 / 256); the
    whole call returns ``None`` if a layout has padding slots (no shape of the ResUNet tables has)."""
    dev = kernels[0][0].device
    key = tuple((tuple(w.shape), bool(m)) for w, m in kernels) + (str(dev),)
    plan = _BULK_CACHE.get(key)
    if plan is None:
        perms, spans, off, pos = [], [], 0, 0
        for w, m in kernels:
            K, cin, cout = w.shape
            n = K * cin * cout
            pf, padded = _pack_perm(K, cin, cout, False, False, dev)
            if padded:
                _BULK_CACHE[key] = False
                return None
            perms.append(pf + off)
            span = [pos, pos + n, None]
            pos += n
            if cin in (32, 64, 128, 256):
                pb, padded = _pack_perm(K, cin, cout, True, m, dev)
                if padded:
                    _BULK_CACHE[key] = False
                    return None
                perms.append(pb + off)
                span[2] = pos + n
                pos += n
            spans.append(span)
            off += n
        plan = (torch.cat(perms).to(torch.int32), spans)            # (index_select takes int32 indices: half the index memory)
        _BULK_CACHE[key] = plan
    if plan is False:
        return None
    perm_all, spans = plan
    with torch.no_grad():
        packed = torch.cat([w.detach().reshape(-1) for w, _ in kernels]).float().index_select(0, perm_all)
    return [(packed[a:b], None if c is None else packed[b:c]) for a, b, c in spans]


def input_gradient(dy: torch.Tensor, weight: torch.Tensor, table_t, mirror: bool, n_in: int, packed_t=None) -> torch.Tensor:
    """d(loss)/d(input) of one layer: the same operator over the transposed rulebook with W[k]^T (mirrored offsets for a self-transposed
    table).  Its output width is the layer's C_in, which the kernels take as 32 / 64 / 128 / 256: the concatenated decoder inputs of some
    channel tables (ResUNetBN2B / BN2D / FatBN: 128 + 64 = 192, 256 + 128 = 384) go through in column blocks of those widths."""
    K, cin, cout = weight.shape
    if cin in (32, 64, 128, 256):
        return _run(table_t, n_in, dy, _pack(weight, True, mirror) if packed_t is None else packed_t, cout, cin)
    if cin % 32:
        raise _lib.EyocError(f"sparse_conv: the input gradient needs C_in % 32 == 0, got {cin}", _lib.ERR_INVALID)
    dx = torch.empty((n_in, cin), dtype=torch.float32, device=dy.device)
    a = 0
    while a < cin:
        w = next(b for b in (256, 128, 64, 32) if b <= cin - a)
        _run(table_t, n_in, dy, _pack(weight[:, a:a + w, :], True, mirror), cout, w, out=dx[:, a:a + w])
        a += w
    return dx


def weight_gradient(x: torch.Tensor, dy: torch.Tensor, weight: torch.Tensor, table) -> torch.Tensor:
    """d(loss)/d(kernel), dense ``[K, C_in, C_out]`` whatever the strides of ``weight`` (``eyoc_spconv_grad_weight``: fp32 MFMA over the
    gathered pair lists, fixed reduction order)."""
    K, cin, cout = weight.shape
    lib = _lib.load()
    n_out = dy.shape[0]
    dw = torch.empty(weight.shape, dtype=torch.float32, device=weight.device)
    with _lib.on_device(x.device):
        ws = _lib.scratch(lib.eyoc_spconv_grad_weight_workspace_bytes(K, n_out, cin, cout), x.device)
        _lib.check(lib.eyoc_spconv_grad_weight(_lib.ctx(x.device.index), _lib.ptr(table), K, n_out, _lib.ptr(x), x.stride(0),
                                               cin, _lib.ptr(dy), dy.stride(0), cout, _lib.ptr(dw), _lib.ptr(ws), ws.numel(),
                                               _lib.stream_ptr()), "eyoc_spconv_grad_weight")
    return dw


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
        dx = input_gradient(dy, weight, ctx.table_t, ctx.mirror, ctx.n_in) if ctx.needs_input_grad[0] else None
        dw = weight_gradient(x, dy, weight, ctx.table) if ctx.needs_input_grad[1] else None
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


def _draw_loss_samples(rng, n0, n1, n_pairs, num_hn_samples, num_pos):
    """The three draws of lib/trainer.py:946-955 in the reference's ORDER (the order is contract: it fixes which numbers a
    seeded ``np.random`` hands out): candidates of cloud 0, candidates of cloud 1, then - only when there are more
    positives than ``num_pos`` - the positive subsample (``None`` = keep all)."""
    cand = [rng.choice(n, min(n, num_hn_samples), replace=False) for n in (n0, n1)]
    keep = rng.choice(n_pairs, num_pos, replace=False) if n_pairs > num_pos else None
    return cand[0], cand[1], keep


def _hardest_negative_term(anchor, cand_feats, cand_ids, anchor_key, cand_key_scale, known_positive_keys, neg_thresh):
    """One direction of the negative term (lib/trainer.py:962-987): for every anchor row the nearest row of
    ``cand_feats`` (``eyoc_knn1`` in "L2" mode, ties to the lowest index), dropped when (anchor, candidate) is itself a
    known positive, penalised by ``relu(neg_thresh - d)^2``.  The distance is re-evaluated with torch ops on the two
    gathered rows so the gradient reaches both ends, exactly where ``D.min(1)`` sends it.

    Everything stays on the device (``cand_ids``, ``anchor_key``, ``known_positive_keys`` - sorted -: int64 device tensors): reading the
    neighbour indices back to filter the positives on the host put two synchronisations between the forward and the backward of
    every iteration; the mean over the kept rows is a masked sum over all of them.  An empty selection gives the reference's NaN
    mean with the reference's gradient (none: a NaN constant is added to a zero sum, instead of dividing by zero)."""
    n = anchor.shape[0]
    with torch.no_grad():
        nearest = knn1_segmented(anchor.detach(), cand_feats.detach(), [0, n], [0, cand_feats.shape[0]], "L2", return_distance=False)
        keys = anchor_key + cand_ids[nearest] * cand_key_scale
        # membership in the SORTED positive keys: one binary search (torch.isin sorts both sides on every call)
        if known_positive_keys.numel():
            at = torch.searchsorted(known_positive_keys, keys).clamp_(max=known_positive_keys.numel() - 1)
            keep = (known_positive_keys[at] != keys).to(anchor.dtype)
        else:
            keep = torch.ones(n, dtype=anchor.dtype, device=anchor.device)
    dist = torch.sqrt((anchor - cand_feats[nearest]).pow(2).sum(1) + 1e-7)
    count = keep.sum()
    empty = torch.where(count > 0, torch.zeros_like(count), torch.full_like(count, float("nan")))
    return (torch.relu(neg_thresh - dist).pow(2) * keep).sum() / count.clamp(min=1.0) + empty


def contrastive_hardest_negative_loss(F0, F1, positive_pairs, num_pos=5192, num_hn_samples=2048, pos_thresh=0.1, neg_thresh=1.4,
                                      rng=None):
    """lib/trainer.py:935-991 -> ``(pos_loss, neg_loss)`` (0-dim tensors with grad).  ``rng`` replaces the reference's
    global ``np.random``.  Pairs are identified by ``i + j * max(N0, N1)`` (util/misc.py:6-18 ``_hash``)."""
    pairs = torch.as_tensor(np.asarray(positive_pairs)) if not isinstance(positive_pairs, torch.Tensor) else positive_pairs
    pairs = pairs.long().cpu()
    scale = max(len(F0), len(F1))
    cand0, cand1, keep = _draw_loss_samples(np.random if rng is None else rng, len(F0), len(F1), len(pairs), num_hn_samples, num_pos)
    used = pairs if keep is None else pairs[torch.as_tensor(keep)]
    dev = F0.device
    # ONE upload of every index array the loss needs (four small synchronous copies from pageable memory otherwise)
    n_used, n_pairs = len(used), len(pairs)
    # (assembled with numpy: a torch.cat of > 32 k elements on the host goes through the intra-op thread pool, and on a box whose cores are
    # capped below the thread count that one call took 13 ms - three times the rest of the loss)
    host = np.concatenate([used.numpy().reshape(-1), pairs.numpy().reshape(-1), np.asarray(cand0, np.int64), np.asarray(cand1, np.int64)])
    devbuf = torch.from_numpy(host).to(dev)
    used_d = devbuf[:2 * n_used].view(n_used, 2)
    pairs_d = devbuf[2 * n_used:2 * (n_used + n_pairs)].view(n_pairs, 2)
    cand0_d = devbuf[2 * (n_used + n_pairs):2 * (n_used + n_pairs) + len(cand0)]
    cand1_d = devbuf[2 * (n_used + n_pairs) + len(cand0):]
    a0, a1 = F0[used_d[:, 0]], F1[used_d[:, 1]]
    all_keys = (pairs_d[:, 0] + pairs_d[:, 1] * scale).sort().values
    # keys: (row of cloud 0) + (row of cloud 1) * scale in both directions
    neg01 = _hardest_negative_term(a0, F1[cand1_d], cand1_d, used_d[:, 0], scale, all_keys, neg_thresh)
    neg10 = _hardest_negative_term(a1, F0[cand0_d], cand0_d, used_d[:, 1] * scale, 1, all_keys, neg_thresh)
    pos = torch.relu((a0 - a1).pow(2).sum(1) - pos_thresh).mean()
    return pos, (neg01 + neg10) / 2
