"""CPU oracle of the FCGF-style sparse ResUNet forward (torch on CPU, fp32).

TEST INFRASTRUCTURE.  **Parity unpinned** (MinkowskiEngine absent, see ``oracle/coords.py``).

Restates ``ResUNet2.forward`` (model/resunet.py:142-193) and ``BasicBlockBase.forward``
(model/residual_block.py:37-53) on top of the output-stationary neighbour tables of
``oracle.coords``.  Each sparse convolution is evaluated the way MinkowskiEngine does it - per
kernel offset: gather the input rows of that offset's pairs, multiply by ``W[k]``, scatter-add into
the output rows - and batch norm is applied un-folded in eval mode (model/common.py:4-6,
``torch.nn.BatchNorm1d`` with eps = 1e-5).
"""
from __future__ import annotations

import numpy as np
import torch

from . import coords as oc

BN_EPS = 1e-5

# class name -> (CHANNELS, TR_CHANNELS), model/resunet.py:12-13,196-231
CHANNEL_TABLES = {
    "ResUNet2": ((None, 32, 64, 128, 256), (None, 32, 64, 64, 128)),
    "ResUNetBN2": ((None, 32, 64, 128, 256), (None, 32, 64, 64, 128)),
    "ResUNetBN2B": ((None, 32, 64, 128, 256), (None, 64, 64, 64, 64)),
    "ResUNetBN2C": ((None, 32, 64, 128, 256), (None, 64, 64, 64, 128)),
    "ResUNetBN2D": ((None, 32, 64, 128, 256), (None, 64, 64, 128, 128)),
    "ResUNetBN2E": ((None, 128, 128, 128, 256), (None, 64, 128, 128, 128)),
    "ResUNetFatBN": ((None, 32, 64, 128, 256), (None, 128, 128, 128, 256)),
    "ResUNetExpBN2C": ((None, 32, 64, 128, 256), (None, 64, 64, 64, 128)),      # model/resunet.py:487-490 (ResUNetExpanded)
}


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.asarray(x))


def sparse_conv(x: torch.Tensor, nbr: np.ndarray, W: torch.Tensor) -> torch.Tensor:
    """``out[o] = sum_k x[nbr[k,o]] @ W[k]`` via per-offset gather -> matmul -> index_add."""
    K, n_out = nbr.shape
    out = torch.zeros((n_out, W.shape[2]), dtype=x.dtype)
    for k in range(K):
        o = np.nonzero(nbr[k] >= 0)[0]
        if len(o) == 0:
            continue
        i = torch.from_numpy(nbr[k][o].astype(np.int64))
        out.index_add_(0, torch.from_numpy(o), x[i] @ W[k])
    return out


_TRAIN = {"on": False, "momentum": 0.1, "running": None, "masks": None}      # set by resunet_forward(train=True) for the duration of a call


def _relu(x, name):
    """ReLU - or, when the caller dictates the decisions (``resunet_forward(relu_masks=...)``), ``x * mask``.  Two fp32
    implementations agree on a pre-activation to ~1e-7 and therefore disagree on the SIGN of the handful that round across
    zero; the value barely moves, the gradient entry flips between g and 0.  A gradient parity test fixes the decisions to the
    product's and checks separately that they differ from this file's own only where the pre-activation is ~0."""
    masks = _TRAIN["masks"]
    if masks is None:
        return torch.relu(x)
    if name is None:                      # a second ReLU on an already rectified tensor (model/resunet.py:146,151,...): identity
        return x
    return x * _t(masks[name]).to(x.dtype)


def batch_norm(x, sd, name):
    """``MinkowskiBatchNorm`` (model/common.py:6 = ``nn.BatchNorm1d`` over the rows); computed in the dtype of ``x``.  Eval mode:
    the running statistics.  Training mode (``resunet_forward(train=True)``): the batch's own mean and biased variance, and the
    running statistics it would leave behind (momentum, unbiased variance) are recorded in ``_TRAIN['running']``."""
    w, b = _t(sd[f"{name}.bn.weight"]).to(x.dtype), _t(sd[f"{name}.bn.bias"]).to(x.dtype)
    m, v = _t(sd[f"{name}.bn.running_mean"]).to(x.dtype), _t(sd[f"{name}.bn.running_var"]).to(x.dtype)
    if _TRAIN["on"]:
        n = x.shape[0]
        mean, var = x.mean(0), x.var(0, unbiased=False)
        if _TRAIN["running"] is not None:
            mom = _TRAIN["momentum"]
            _TRAIN["running"][f"{name}.bn.running_mean"] = ((1 - mom) * m + mom * mean).detach()
            _TRAIN["running"][f"{name}.bn.running_var"] = ((1 - mom) * v + mom * var * n / max(n - 1, 1)).detach()
        m, v = mean, var
    return (x - m) / torch.sqrt(v + BN_EPS) * w + b


def _kernel(sd, name, dtype=torch.float32):
    w = _t(sd[f"{name}.kernel"]).to(dtype)
    return w[None] if w.dim() == 2 else w       # 1x1 convs store a 2-D kernel


def basic_block(x, nbr, sd, name, stored=None):
    """model/residual_block.py:37-53 (downsample is always None, model/resunet.py:41-42).  ``stored``: optional dict that
    receives the two tensors a fused implementation materialises (after conv1 + norm1 + ReLU, after the residual ReLU)."""
    out = sparse_conv(x, nbr, _kernel(sd, f"{name}.conv1", x.dtype))
    out = _relu(batch_norm(out, sd, f"{name}.norm1"), f"{name}.conv1")
    if stored is not None:
        stored[f"{name}.conv1"] = out
    out = sparse_conv(out, nbr, _kernel(sd, f"{name}.conv2", x.dtype))
    out = batch_norm(out, sd, f"{name}.norm2")
    out = _relu(out + x, f"{name}.conv2")
    if stored is not None:
        stored[f"{name}.conv2"] = out
    return out


def resunet_forward(sd: dict, coords: np.ndarray, feats, normalize_feature=True,
                    conv1_kernel_size=5, maps=None, return_intermediate=False, dtype=torch.float32, train=False,
                    bn_momentum=0.1, running_out=None, relu_masks=None):
    """Forward of ``ResUNet2`` (any BN channel table - shapes come from ``sd``).

    ``coords int [N,4] (b,x,y,z)``, ``feats f32 [N,C_in]`` -> ``[N,C_out]`` in input row order.  ``dtype``: the
    arithmetic (fp32 like the reference; ``torch.float64`` gives the error yardstick the split16 tests use).
    ``train=True``: every batch norm uses batch statistics (``model.train()``, lib/trainer.py:1655-1676); entries of ``sd`` that
    are tensors with ``requires_grad`` get gradients from ``out.backward()``; ``running_out`` (a dict) receives the running
    statistics a ``bn_momentum`` update leaves behind.  ``relu_masks``: ``{stored-layer name: 0/1 tensor}`` - the ReLU decisions
    to use instead of this forward's own (see ``_relu``); names as in ``inter["stored"]`` (``block1.conv1``, ``block1.conv2`` ...,
    ``conv1_tr``).
    """
    _TRAIN.update(on=bool(train), momentum=bn_momentum, running=running_out, masks=relu_masks)
    try:
        return _resunet_forward(sd, coords, feats, normalize_feature, conv1_kernel_size, maps, return_intermediate, dtype)
    finally:
        _TRAIN.update(on=False, running=None, masks=None)


def _second(out, nbr, sd, i, stored):
    """``ResUNetExpanded`` (model/resunet.py:426-471): ``relu -> norm<i>_2 -> block<i>_2`` behind a stage's block, when the state
    dict has those layers."""
    if f"norm{i}_2.bn.weight" not in sd:
        return out
    return basic_block(batch_norm(_relu(out, None), sd, f"norm{i}_2"), nbr, sd, f"block{i}_2", stored)


def _resunet_forward(sd, coords, feats, normalize_feature, conv1_kernel_size, maps, return_intermediate, dtype):
    if maps is None:
        maps = oc.build_maps(coords, conv1_kernel_size)
    s1, down, up = maps["s1"], maps["down"], maps["up"]
    ident = lambda n: np.arange(n, dtype=np.int32)[None, :]
    x = _t(feats).to(dtype)
    inter = {}
    # per conv layer (state_dict name), the tensor an implementation that fuses norm / ReLU / residual into the convolution
    # keeps in memory after it - what the split16 range-guard tests compare the product's activation probe with
    stored = {} if return_intermediate else None
    _kernel = lambda sd_, name: globals()["_kernel"](sd_, name, dtype)

    def keep(name, t):
        if stored is not None:
            stored[name] = t
        return t

    # encoder (model/resunet.py:143-161)
    out_s1 = keep("conv1", batch_norm(sparse_conv(x, maps["k5"], _kernel(sd, "conv1")), sd, "norm1"))
    out_s1 = basic_block(out_s1, s1[0], sd, "block1", stored)
    out_s1 = _second(out_s1, s1[0], sd, "1", stored)
    out = _relu(out_s1, None)
    out_s2 = keep("conv2", batch_norm(sparse_conv(out, down[0], _kernel(sd, "conv2")), sd, "norm2"))
    out_s2 = basic_block(out_s2, s1[1], sd, "block2", stored)
    out_s2 = _second(out_s2, s1[1], sd, "2", stored)
    out = _relu(out_s2, None)
    out_s4 = keep("conv3", batch_norm(sparse_conv(out, down[1], _kernel(sd, "conv3")), sd, "norm3"))
    out_s4 = basic_block(out_s4, s1[2], sd, "block3", stored)
    out_s4 = _second(out_s4, s1[2], sd, "3", stored)
    out = _relu(out_s4, None)
    out_s8 = keep("conv4", batch_norm(sparse_conv(out, down[2], _kernel(sd, "conv4")), sd, "norm4"))
    out_s8 = basic_block(out_s8, s1[3], sd, "block4", stored)
    out_s8 = _second(out_s8, s1[3], sd, "4", stored)
    out = _relu(out_s8, None)
    inter.update(out_s1=out_s1, out_s2=out_s2, out_s4=out_s4, out_s8=out_s8)

    # decoder (model/resunet.py:163-186); ME.cat order is [decoder | skip]
    out = keep("conv4_tr", batch_norm(sparse_conv(out, up[2], _kernel(sd, "conv4_tr")), sd, "norm4_tr"))
    out_s4_tr = _relu(_second(basic_block(out, s1[2], sd, "block4_tr", stored), s1[2], sd, "4_tr", stored), None)
    out = torch.cat([out_s4_tr, out_s4], 1)
    out = keep("conv3_tr", batch_norm(sparse_conv(out, up[1], _kernel(sd, "conv3_tr")), sd, "norm3_tr"))
    out_s2_tr = _relu(_second(basic_block(out, s1[1], sd, "block3_tr", stored), s1[1], sd, "3_tr", stored), None)
    out = torch.cat([out_s2_tr, out_s2], 1)
    out = keep("conv2_tr", batch_norm(sparse_conv(out, up[0], _kernel(sd, "conv2_tr")), sd, "norm2_tr"))
    out_s1_tr = _relu(_second(basic_block(out, s1[0], sd, "block2_tr", stored), s1[0], sd, "2_tr", stored), None)
    out = torch.cat([out_s1_tr, out_s1], 1)
    inter.update(out_s4_tr=out_s4_tr, out_s2_tr=out_s2_tr, out_s1_tr=out_s1_tr)
    n = out.shape[0]
    out = keep("conv1_tr", _relu(sparse_conv(out, ident(n), _kernel(sd, "conv1_tr")), "conv1_tr"))
    out = sparse_conv(out, ident(n), _kernel(sd, "final")) + _t(sd["final.bias"]).to(dtype).reshape(1, -1)
    inter["stored"] = stored
    inter["pre_norm"] = out
    if normalize_feature:
        # model/resunet.py:187-191 - no epsilon: a zero row yields NaN, as in the reference
        out = out / torch.norm(out, p=2, dim=1, keepdim=True)
    if return_intermediate:
        return out, inter, maps
    return out


def work_model(stats: dict, sd: dict) -> dict:
    """Algorithmic FLOPs / bytes of one forward from realised map sizes (SURVEY.md §8d formulas):
    ``FLOP = 2 pairs Cin Cout``; ``gather_bytes = pairs (4 Cin + 8) + 4 N_out Cout + 4 K Cin Cout``;
    ``compulsory_bytes = 4 (N_in Cin + N_out Cout) + 8 pairs + 4 K Cin Cout``."""
    n = stats["rows"]
    layers = []

    def add(name, pairs, n_in, n_out):
        w = _kernel(sd, name)
        K, ci, co = w.shape
        layers.append({
            "name": name, "pairs": pairs, "flop": 2 * pairs * ci * co,
            "gather_bytes": pairs * (4 * ci + 8) + 4 * n_out * co + 4 * K * ci * co,
            "compulsory_bytes": 4 * (n_in * ci + n_out * co) + 8 * pairs + 4 * K * ci * co,
        })

    add("conv1", stats["pairs_k5"], n[0], n[0])
    for lvl, blk in ((0, "block1"), (1, "block2"), (2, "block3"), (3, "block4"),
                     (2, "block4_tr"), (1, "block3_tr"), (0, "block2_tr")):
        for c in ("conv1", "conv2"):
            add(f"{blk}.{c}", stats["pairs_s1"][lvl], n[lvl], n[lvl])
    for i, name in enumerate(("conv2", "conv3", "conv4")):
        add(name, stats["pairs_down"][i], n[i], n[i + 1])
    for i, name in ((2, "conv4_tr"), (1, "conv3_tr"), (0, "conv2_tr")):
        add(name, stats["pairs_up"][i], n[i + 1], n[i])
    add("conv1_tr", n[0], n[0], n[0])
    add("final", n[0], n[0], n[0])
    tot = {k: sum(l[k] for l in layers) for k in ("flop", "gather_bytes", "compulsory_bytes")}
    return {"layers": layers, **tot}
