"""Training-mode forward of ``ResUNet2`` with autograd.

``lib/trainer.py:1655-1676`` runs ``model(sinput)`` in train mode and ``loss.backward()`` through MinkowskiEngine.  Here
``model.train()(x)`` returns a ``SparseTensor`` whose ``.F`` carries a graph of ``torch.autograd.Function`` nodes that run in
``libeyoc_hip.so``:

  * every sparse convolution = ``autograd.sparse_conv`` (fp32 MFMA forward; grad-input = the forward kernel over the transposed
    rulebook; grad-weight = ``eyoc_spconv_grad_weight``);
  * the first convolution (C_in = 1 in production: nothing for the 32-channel-block kernels to chew on) = the 5^3 window
    gathered into a dense ``[N, 125]`` matrix by ``eyoc_maps_gather_window`` and one plain product with the ``[125, 32]``
    kernel (a library GEMM; its weight gradient is the transposed product);
  * batch normalisation with batch statistics = ``eyoc_bn_train_forward / _backward`` (fp64 sums in a fixed order), the ReLU that
    follows a norm fused into it; running statistics updated like ``nn.BatchNorm1d`` (momentum, unbiased variance);
  * the two 1x1 layers at the end are plain ``[N, C] x [C, C']`` products (library GEMMs); residual adds, concatenations and
    the final row normalisation are element-wise torch ops.

Same layer graph as the eval forward (model/resunet.py:142-193, model/residual_block.py:37-53); the maps are built in the
caller's row order.  Eval-mode inference does not come through here (it runs the fused, folded kernels of ``model.hip``)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from . import autograd as _ag
from .autograd import sparse_conv
from .sparse_tensor import SparseTensor

MAP_S1, MAP_DOWN, MAP_UP = 0, 1, 2


class _BatchNormTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, running=None):
        """``running = (running_mean, running_var, momentum)``: moved inside the same library call (one launch instead of five
        element-wise ones per norm); ``None``: statistics only."""
        x = x.contiguous()
        n, c = x.shape
        lib = _lib.load()
        y = torch.empty_like(x)
        stats = torch.empty(2 * c, dtype=torch.float32, device=x.device)
        with _lib.on_device(x.device):
            ws = _lib.scratch(lib.eyoc_bn_workspace_bytes(n, c), x.device)
            if running is not None:
                _lib.check(lib.eyoc_bn_train_forward_running(_lib.ctx(x.device.index), _lib.ptr(x), n, c, x.stride(0), _lib.ptr(gamma.contiguous()),
                                                             _lib.ptr(beta.contiguous()), float(eps), 1 if relu else 0, _lib.ptr(y), y.stride(0),
                                                             _lib.ptr(stats), _lib.ptr(running[0]), _lib.ptr(running[1]), float(running[2]),
                                                             _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "eyoc_bn_train_forward_running")
            else:
                _lib.check(lib.eyoc_bn_train_forward(_lib.ctx(x.device.index), _lib.ptr(x), n, c, x.stride(0), _lib.ptr(gamma.contiguous()),
                                                     _lib.ptr(beta.contiguous()), float(eps), 1 if relu else 0, _lib.ptr(y), y.stride(0),
                                                     _lib.ptr(stats), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "eyoc_bn_train_forward")
        ctx.save_for_backward(x, y if relu else None, gamma, stats)
        ctx.eps, ctx.relu = float(eps), bool(relu)
        ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        x, y, gamma, stats = ctx.saved_tensors
        dy = dy.contiguous()
        n, c = x.shape
        lib = _lib.load()
        dx = torch.empty_like(x)
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
        with _lib.on_device(x.device):
            ws = _lib.scratch(lib.eyoc_bn_workspace_bytes(n, c), x.device)
            _lib.check(lib.eyoc_bn_train_backward(_lib.ctx(x.device.index), _lib.ptr(x), x.stride(0), _lib.ptr(y), 0 if y is None else y.stride(0),
                                                  _lib.ptr(dy), dy.stride(0), n, c, _lib.ptr(gamma.contiguous()), _lib.ptr(stats), ctx.eps,
                                                  _lib.ptr(dx), dx.stride(0), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ws), ws.numel(),
                                                  _lib.stream_ptr()), "eyoc_bn_train_backward")
        return dx, dgamma, dbeta, None, None, None


class _ConvNormTrain(torch.autograd.Function):
    """One sparse convolution and the batch norm (+ ReLU) behind it as ONE autograd node: the two library calls of ``sparse_conv`` and
    ``_BatchNormTrain`` back to back, forward and backward - every convolution of the network is followed by a norm, and a training
    iteration on two 11 k-voxel clouds is bound by the ~20 us of host time an autograd node costs per direction."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, table, table_t, mirror, n_out, eps, relu, running, packed):
        x = x.contiguous()
        K, cin, cout = weight.shape
        z = _ag._run(table, n_out, x, _ag._pack(weight, False, False) if packed is None else packed[0], cin, cout)
        ctx.packed_t = None if packed is None else packed[1]
        lib = _lib.load()
        y = torch.empty_like(z)
        stats = torch.empty(2 * cout, dtype=torch.float32, device=x.device)
        with _lib.on_device(x.device):
            ws = _lib.scratch(lib.eyoc_bn_workspace_bytes(n_out, cout), x.device)
            _lib.check(lib.eyoc_bn_train_forward_running(_lib.ctx(x.device.index), _lib.ptr(z), n_out, cout, z.stride(0), _lib.ptr(gamma),
                                                         _lib.ptr(beta), float(eps), 1 if relu else 0, _lib.ptr(y), y.stride(0),
                                                         _lib.ptr(stats), _lib.ptr(running[0]), _lib.ptr(running[1]), float(running[2]),
                                                         _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "eyoc_bn_train_forward_running")
        ctx.save_for_backward(x, weight, z, y if relu else None, gamma, stats)
        ctx.table, ctx.table_t, ctx.mirror, ctx.n_in, ctx.eps = table, table_t, mirror, x.shape[0], float(eps)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, z, y, gamma, stats = ctx.saved_tensors
        dy = dy.contiguous()
        n, c = z.shape
        lib = _lib.load()
        dz = torch.empty_like(z)
        dgamma = torch.empty(c, dtype=torch.float32, device=z.device)
        dbeta = torch.empty(c, dtype=torch.float32, device=z.device)
        with _lib.on_device(z.device):
            ws = _lib.scratch(lib.eyoc_bn_workspace_bytes(n, c), z.device)
            _lib.check(lib.eyoc_bn_train_backward(_lib.ctx(z.device.index), _lib.ptr(z), z.stride(0), _lib.ptr(y), 0 if y is None else y.stride(0),
                                                  _lib.ptr(dy), dy.stride(0), n, c, _lib.ptr(gamma), _lib.ptr(stats), ctx.eps,
                                                  _lib.ptr(dz), dz.stride(0), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ws), ws.numel(),
                                                  _lib.stream_ptr()), "eyoc_bn_train_backward")
        dx = _ag.input_gradient(dz, weight, ctx.table_t, ctx.mirror, ctx.n_in, ctx.packed_t) if ctx.needs_input_grad[0] else None
        dw = _ag.weight_gradient(x, dz, weight, ctx.table) if ctx.needs_input_grad[1] else None
        return dx, dw, dgamma, dbeta, None, None, None, None, None, None, None, None


def conv_norm_train(x: torch.Tensor, weight: torch.Tensor, bn: torch.nn.BatchNorm1d, table, table_t=None, relu: bool = False,
                    packed=None) -> torch.Tensor:
    """``norm(conv(x))`` in training mode; the fused node when the norm is an ordinary tracking fp32 ``BatchNorm1d``, the two separate
    ones otherwise (same arithmetic either way: the same two library calls)."""
    fused = bn.track_running_stats and bn.running_mean is not None and bn.momentum is not None and bn.weight is not None \
        and bn.bias is not None and all(t.dtype == torch.float32 and t.is_contiguous() and t.device == x.device
                                        for t in (bn.running_mean, bn.running_var, bn.weight, bn.bias)) and table is not None
    if not fused:
        return batch_norm_train(sparse_conv(x, weight, table, table_t), bn, relu)
    mirror = table_t is None
    y = _ConvNormTrain.apply(x, weight, bn.weight, bn.bias, table, table if mirror else table_t, mirror, table.shape[1], bn.eps, relu,
                             (bn.running_mean, bn.running_var, bn.momentum), packed)
    _touch_running(bn)
    return y


def _touch_running(bn):
    """The library moved ``running_mean`` / ``running_var`` through raw pointers: tell torch (``_version`` is what autograd's saved-tensor
    check and ``model._weights_version()`` look at) and count the batch like ``nn.BatchNorm1d``."""
    torch.autograd.graph.increment_version((bn.running_mean, bn.running_var))
    if bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1


def batch_norm_train(x: torch.Tensor, bn: torch.nn.BatchNorm1d, relu: bool = False) -> torch.Tensor:
    """``MinkowskiBatchNorm`` in training mode (model/common.py:4-6): normalise with the batch's own statistics and move
    the running statistics by ``momentum`` (unbiased variance, ``num_batches_tracked + 1``) exactly like ``nn.BatchNorm1d``."""
    track = bn.track_running_stats and bn.running_mean is not None
    if track and bn.momentum is not None and bn.running_mean.is_contiguous() and bn.running_var.is_contiguous() \
            and bn.running_mean.dtype == torch.float32:
        y, _ = _BatchNormTrain.apply(x, bn.weight, bn.bias, bn.eps, relu, (bn.running_mean, bn.running_var, bn.momentum))
        _touch_running(bn)
        return y
    y, stats = _BatchNormTrain.apply(x, bn.weight, bn.bias, bn.eps, relu)
    if track:
        n, c = x.shape
        with torch.no_grad():
            bn.num_batches_tracked += 1
            m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            mean, var = stats[:c], stats[c:]
            bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
            bn.running_var.mul_(1 - m).add_(var * (n / max(n - 1, 1)), alpha=m)
    return y


def gather_window(cm, feats: torch.Tensor, ks: int, internal: bool = False) -> torch.Tensor:
    """``[N, ks^3 * C_in]``: every row's ``ks^3`` window of input features (zeros where no voxel is), caller's row order
    (``internal=True``: ``feats`` and the result in the rows of the forward's own - possibly Z-ordered - maps)."""
    n, cin = feats.shape
    out = torch.empty((n, ks ** 3 * cin), dtype=torch.float32, device=feats.device)
    lib = _lib.load()
    with _lib.on_device(feats.device):
        if internal:
            _lib.check(lib.eyoc_maps_gather_window_internal(_lib.ctx(feats.device.index), cm.maps(), int(ks), _lib.ptr(feats.contiguous()), cin,
                                                            _lib.ptr(out), _lib.stream_ptr()), "eyoc_maps_gather_window_internal")
        else:
            _lib.check(lib.eyoc_maps_gather_window(_lib.ctx(feats.device.index), cm._caller_maps(), int(ks), _lib.ptr(feats.contiguous()), cin,
                                                   _lib.ptr(out), _lib.stream_ptr()), "eyoc_maps_gather_window")
    return out


def _match_rows(want: torch.Tensor, have: torch.Tensor) -> torch.Tensor:
    """For two orderings of the same unique coordinate rows ``[n, 4]``: ``idx`` with ``have[idx] == want`` (diagnostics only)."""
    def key(c):
        c = c.long()
        return ((c[:, 0] * (1 << 18) + (c[:, 1] + (1 << 17))) * (1 << 18) + (c[:, 2] + (1 << 17))) * (1 << 18) + (c[:, 3] + (1 << 17))
    kw, kh = key(want), key(have)
    sh = torch.argsort(kh)
    return sh[torch.searchsorted(kh[sh], kw)]


def forward_train(model, x: SparseTensor, taps: dict | None = None) -> SparseTensor:
    """model/resunet.py:142-193 with batch statistics; every parameter of ``model`` receives a gradient from ``.backward()``.
    ``taps`` (tests / diagnostics): receives every rectified tensor under its layer's name (``block1.conv1`` = after the
    block's first norm + ReLU, ``block1.conv2`` = the block's output, ..., ``conv1_tr``) - the ReLU decisions of this forward."""
    return forward_layers(model, x, taps)


def forward_layers(model, x: SparseTensor, taps: dict | None = None) -> SparseTensor:
    """The network layer by layer through the autograd Functions above - ``ResUNet2`` (model/resunet.py:142-193) and
    ``ResUNetExpanded`` (:254-484: every stage runs a second norm + block, ``norm<i>_2`` / ``block<i>_2``).  In training mode
    every norm uses batch statistics; in eval mode its running statistics (a per-channel affine, element-wise) - ``model(x)`` in eval
    mode does not come here (it runs the packed plan, csrc/model.hip); tests call this under ``no_grad`` to have the same network twice."""
    cm = x.coordinate_manager
    # The layers run in the rows of the forward's own maps: from 8192 rows on those are Z-ordered (eyoc_maps_build_ordered(-1): one
    # sort + top-down derivation, two host synchronisations) - a second set of maps in the caller's order is the hash-table build with a
    # host round trip per level, 2.1 ms of a 11 ms training iteration on two 11 k-voxel clouds.  Input rows are permuted on the way in,
    # output rows (and the tapped tensors) on the way out; both are differentiable row gathers.
    cm.maps(-1)
    order = cm.row_order()                                   # int32 [N]: caller's row of internal row i; None = the caller's order was kept
    internal = order is not None
    s1 = [cm.table_view(MAP_S1, l, internal=internal) for l in range(4)]
    down = [cm.table_view(MAP_DOWN, l, internal=internal) for l in range(3)]
    up = [cm.table_view(MAP_UP, l, internal=internal) for l in range(3)]
    expanded = bool(getattr(model, "EXPANDED", False))
    back = None
    if internal:
        order = order.long()
        back = torch.empty_like(order)
        back[order] = torch.arange(order.numel(), device=order.device)      # internal row of every caller's row
    level_back = {}

    def tap(name, t, lvl):
        """``lvl``: the tensor's level, from the call site (two levels can have the same number of rows - isolated voxels that a
        strided convolution does not merge - so the row count does not identify it)"""
        if taps is not None:
            if internal:                                     # diagnostics see the caller's rows at every level
                assert t.shape[0] == cm.rows(lvl), (name, t.shape[0], lvl, cm.rows(lvl))
                if lvl not in level_back:
                    zc, cc = cm.level_coordinates(lvl, internal=True), cm.level_coordinates(lvl)
                    level_back[lvl] = _match_rows(cc, zc)
                taps[name] = t.index_select(0, level_back[lvl])
            else:
                taps[name] = t
        return t

    def norm(t, n, relu=False):
        if model.training:
            return batch_norm_train(t, n.bn, relu)
        bn = n.bn
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        y = t * scale + (bn.bias - bn.running_mean * scale)
        return torch.relu(y) if relu else y

    # every kernel of the fused nodes packed (forward + transposed) by one gather
    packs = {}
    if model.training:
        stages = ("1", "2", "3", "4", "4_tr", "3_tr", "2_tr")
        blocks = [getattr(model, "block" + n) for n in stages] + ([getattr(model, f"block{n}_2") for n in stages] if expanded else [])
        kernels = [(getattr(model, "conv" + n).kernel, False) for n in stages[1:]]
        kernels += [(c.kernel, True) for b in blocks for c in (b.conv1, b.conv2)]
        bulk = _ag.bulk_pack(kernels)
        if bulk is not None:
            packs = {id(w): pk for (w, _), pk in zip(kernels, bulk)}

    def conv_norm(t, kernel, n, table, table_t=None, relu=False):
        """a convolution and the norm behind it (one autograd node in training mode)"""
        if model.training:
            return conv_norm_train(t, kernel, n.bn, table, table_t, relu, packs.get(id(kernel)))
        return norm(sparse_conv(t, kernel, table, table_t), n, relu)

    def block(t, blk, table, name, lvl):
        """BasicBlockBN (model/residual_block.py:37-53): relu(bn2(conv2(relu(bn1(conv1(x))))) + x)"""
        out = tap(name + ".conv1", conv_norm(t, blk.conv1.kernel, blk.norm1, table, relu=True), lvl)
        out = conv_norm(out, blk.conv2.kernel, blk.norm2, table)
        return tap(name + ".conv2", torch.relu(out + t), lvl)

    def stage(t, name, lvl, conv=None):
        """``[conv ->] norm -> block (-> relu, already rectified) [-> norm_2 -> block_2]``; ``conv = (input, kernel, table, table_t)`` is
        the stage's own convolution (fused with the stage's norm), ``t`` its output when the caller ran it already"""
        table = s1[lvl]
        first = conv_norm(conv[0], conv[1], getattr(model, "norm" + name), conv[2], conv[3]) if conv else norm(t, getattr(model, "norm" + name))
        out = block(first, getattr(model, "block" + name), table, "block" + name, lvl)
        if expanded:
            out = block(norm(out, getattr(model, f"norm{name}_2")), getattr(model, f"block{name}_2"), table, f"block{name}_2", lvl)
        return out

    # encoder.  conv1: window gather + one dense product (C_in is tiny)
    F_in = x.F.index_select(0, order) if internal else x.F
    G = gather_window(cm, F_in, model.conv1_kernel_size, internal=internal)
    out_s1 = stage(G @ model.conv1.kernel.reshape(-1, model.conv1.cout), "1", 0)
    out_s2 = stage(None, "2", 1, (out_s1, model.conv2.kernel, down[0], up[0]))
    out_s4 = stage(None, "3", 2, (out_s2, model.conv3.kernel, down[1], up[1]))
    out_s8 = stage(None, "4", 3, (out_s4, model.conv4.kernel, down[2], up[2]))
    # decoder; ME.cat order is [decoder | skip]
    out = stage(None, "4_tr", 2, (out_s8, model.conv4_tr.kernel, up[2], down[2]))
    out = torch.cat([out, out_s4], 1)
    out = stage(None, "3_tr", 1, (out, model.conv3_tr.kernel, up[1], down[1]))
    out = torch.cat([out, out_s2], 1)
    out = stage(None, "2_tr", 0, (out, model.conv2_tr.kernel, up[0], down[0]))
    out = torch.cat([out, out_s1], 1)
    # the two 1x1 layers are plain dense products (96 -> 64 -> 32): library GEMMs, forward and backward
    out = tap("conv1_tr", torch.relu(out @ model.conv1_tr.kernel), 0)
    out = out @ model.final.kernel + model.final.bias
    if model.normalize_feature:
        out = out / torch.norm(out, p=2, dim=1, keepdim=True)          # no epsilon (model/resunet.py:187-191)
    if internal:
        out = out.index_select(0, back)
    return SparseTensor(out, coordinate_map_key=x.coordinate_map_key, coordinate_manager=cm)
