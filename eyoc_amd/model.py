"""ResUNet2 family and ``load_model`` with the reference's constructor / ``__call__`` surface.

Mirrors ``model/resunet.py:10-251`` (``ResUNet2`` and its BN channel-table subclasses),
``model/residual_block.py:9-61`` (parameter names of ``BasicBlockBN``), ``model/common.py:4-6`` and the
name registry of ``model/__init__.py:8-30``.  Parameters carry MinkowskiEngine's ``state_dict`` names
(``conv1.kernel [K,Cin,Cout]``, ``norm1.bn.weight``, ``block1.conv1.kernel``, ``final.bias [1,Cout]`` ...)
so ``model.load_state_dict(torch.load(...)['state_dict'])`` works as in ``scripts/test_kitti.py:90-91``.

The eval-mode forward (the path EYOC's test / labelling code uses) runs entirely in ``libeyoc_hip.so``: the parameters
are folded (BN into conv) and packed once per weight version into a single device blob, which is also what
``eyoc_amd.dist`` broadcasts between ranks.  In training mode ``model(x)`` is a differentiable layer-by-layer forward with
batch statistics (``eyoc_amd/train.py``).
"""
from __future__ import annotations

import ctypes as C
import itertools
import logging

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .sparse_tensor import SparseTensor

_TENSOR_VERSION = torch.Tensor._version.__get__      # the getters as plain C callables (``_weights_version`` maps them over the tensors)
_TENSOR_PTR = torch.Tensor.data_ptr


class _Conv(nn.Module):
    """Parameter holder named like ``ME.MinkowskiConvolution``: ``kernel`` (+ ``bias [1,C]``)."""

    def __init__(self, K, cin, cout, bias=False):
        super().__init__()
        shape = (cin, cout) if K == 1 else (K, cin, cout)
        self.kernel = nn.Parameter(torch.randn(shape) * float(np.sqrt(2.0 / (K * cin))))
        self.bias = nn.Parameter(torch.zeros(1, cout)) if bias else None
        self.K, self.cin, self.cout = K, cin, cout


class _Norm(nn.Module):
    """``ME.MinkowskiBatchNorm`` keeps its ``nn.BatchNorm1d`` under ``.bn`` (model/common.py:6)."""

    def __init__(self, c, momentum):
        super().__init__()
        self.bn = nn.BatchNorm1d(c, momentum=momentum)


class _Block(nn.Module):
    """``BasicBlockBN`` (model/residual_block.py:9-35): conv1, norm1, conv2, norm2."""

    def __init__(self, c, momentum):
        super().__init__()
        self.conv1 = _Conv(27, c, c)
        self.norm1 = _Norm(c, momentum)
        self.conv2 = _Conv(27, c, c)
        self.norm2 = _Norm(c, momentum)


class ResUNet2(nn.Module):
    EXPANDED = False
    NORM_TYPE = None
    BLOCK_NORM_TYPE = 'BN'
    CHANNELS = [None, 32, 64, 128, 256]
    TR_CHANNELS = [None, 32, 64, 64, 128]

    def __init__(self, in_channels=3, out_channels=32, bn_momentum=0.1, normalize_feature=None,
                 conv1_kernel_size=None, D=3):
        super().__init__()
        if D != 3:
            raise ValueError("only D=3 is supported")
        if self.NORM_TYPE != 'BN' or self.BLOCK_NORM_TYPE != 'BN':
            # ResUNet2 itself has NORM_TYPE None (get_norm would raise in the reference too);
            # the IN variants need instance norm, which is outside the hot path
            raise NotImplementedError(f"{type(self).__name__}: only batch-norm variants are implemented")
        if conv1_kernel_size is None:
            raise ValueError("conv1_kernel_size is required (config.py:84 default is 5)")
        Cn, T = self.CHANNELS, self.TR_CHANNELS
        self.in_channels, self.out_channels = in_channels, out_channels
        self.conv1_kernel_size = conv1_kernel_size
        self.normalize_feature = normalize_feature
        m = bn_momentum
        self.conv1 = _Conv(conv1_kernel_size ** 3, in_channels, Cn[1]); self.norm1 = _Norm(Cn[1], m)
        self.block1 = _Block(Cn[1], m)
        self.conv2 = _Conv(27, Cn[1], Cn[2]); self.norm2 = _Norm(Cn[2], m); self.block2 = _Block(Cn[2], m)
        self.conv3 = _Conv(27, Cn[2], Cn[3]); self.norm3 = _Norm(Cn[3], m); self.block3 = _Block(Cn[3], m)
        self.conv4 = _Conv(27, Cn[3], Cn[4]); self.norm4 = _Norm(Cn[4], m); self.block4 = _Block(Cn[4], m)
        self.conv4_tr = _Conv(27, Cn[4], T[4]); self.norm4_tr = _Norm(T[4], m); self.block4_tr = _Block(T[4], m)
        self.conv3_tr = _Conv(27, Cn[3] + T[4], T[3]); self.norm3_tr = _Norm(T[3], m); self.block3_tr = _Block(T[3], m)
        self.conv2_tr = _Conv(27, Cn[2] + T[3], T[2]); self.norm2_tr = _Norm(T[2], m); self.block2_tr = _Block(T[2], m)
        if self.EXPANDED:            # ResUNetExpanded (model/resunet.py:254-425): a second norm + block per stage
            for i, c in (("1", Cn[1]), ("2", Cn[2]), ("3", Cn[3]), ("4", Cn[4]), ("4_tr", T[4]), ("3_tr", T[3]), ("2_tr", T[2])):
                setattr(self, f"norm{i}_2", _Norm(c, m))
                setattr(self, f"block{i}_2", _Block(c, m))
        self.conv1_tr = _Conv(1, Cn[1] + T[2], T[1])
        self.final = _Conv(1, T[1], out_channels, bias=True)
        self._handle = None      # eyoc_model*
        self._blob = None        # packed weights (device tensor, float32)
        self._packed_device = None
        self._packed_version = None
        self._timing = False
        self._math = -1          # sparse-conv arithmetic: -1 automatic, 0 fp32 MFMA, 1 split16 (see spconv_math)
        # split16 range guard: True = every split16 forward is followed by eyoc_model_range_check (one stream
        # synchronisation); pipelined callers set it False and call check_range() where they synchronise anyway
        self.range_check = True
        self._probe = False
        self.last_max_activation = None

    # ------------------------------------------------------------------ packing
    def _desc(self):
        d = _lib.ModelDesc()
        d.in_channels, d.out_channels = self.in_channels, self.out_channels
        d.conv1_kernel_size = self.conv1_kernel_size
        d.normalize_feature = 1 if self.normalize_feature else 0
        for i in range(1, 5):
            d.channels[i], d.tr_channels[i] = self.CHANNELS[i], self.TR_CHANNELS[i]
        d.bn_eps = 1e-5
        d.expanded = 1 if self.EXPANDED else 0
        return d

    def _weights_version(self):
        """Fingerprint of (storage, autograd version counter) of every parameter / buffer: it changes when a tensor is
        rebound (``load_state_dict``, ``.to()``) or modified in place THROUGH the tensor itself (``p.copy_()``,
        ``p.mul_()`` under ``no_grad`` - the EMA labeler sync of lib/trainer.py:1509-1513 works that way).  It does NOT
        see edits made through ``p.data`` (``p.data.mul_()`` leaves ``p._version`` alone) nor writes through another
        view of the storage: after those call ``repack()`` - the forward would silently run the old packed weights.
        Inference-mode tensors have no version counter; they contribute their storage address only."""
        # walked by hand: ``self.parameters()`` / ``self.buffers()`` go through torch's generic named-member generators - 0.3 ms per
        # forward for the 73 modules of ResUNetBN2C, a fifth of a single pair's 1.6 ms.  The module list is cached together with every
        # module's children, which are re-checked here (a replaced sub-module anywhere in the tree rebuilds the list); parameters and
        # buffers are read from the modules' own dicts every time, so rebinding one (``.to()``, ``register_buffer``) is seen.
        # Round 6: the walk itself runs in C where it can (``map`` over the tensors with the unbound ``data_ptr`` / ``_version`` getters,
        # the modules' child tuples compared as one list): 128 -> 67 us per forward, a single pair's call is 1.5 ms.
        cache = self.__dict__.get("_eyoc_module_list")
        if cache is None or [tuple(d.values()) for d in cache[0]] != cache[1]:      # (modules compare by identity)
            mods = list(self.modules())
            kid_dicts = [m._modules for m in mods]
            cache = (kid_dicts, [tuple(d.values()) for d in kid_dicts], [d for m in mods for d in (m._parameters, m._buffers)])
            self.__dict__["_eyoc_module_list"] = cache
        ts = [t for t in itertools.chain.from_iterable(map(dict.values, cache[2])) if t is not None]
        try:
            vers = tuple(map(_TENSOR_VERSION, ts))
        except RuntimeError:                                                          # inference-mode tensors have no version counter
            vers = []
            for t in ts:
                try:
                    vers.append(t._version)
                except RuntimeError:
                    vers.append(-1)
            vers = tuple(vers)
        return hash((tuple(map(_TENSOR_PTR, ts)), vers))

    def _invalidate(self):
        if self._handle is not None:
            _lib.load().eyoc_model_destroy(self._handle)
        self._handle, self._blob, self._packed_device = None, None, None

    def load_state_dict(self, state_dict, strict=True):
        out = super().load_state_dict(state_dict, strict=strict)
        self._invalidate()
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._invalidate()
        return out

    def blob_floats(self) -> int:
        d = self._desc()
        return int(_lib.load().eyoc_model_blob_floats(C.byref(d)))

    def _layer_params(self):
        """Host-side (numpy) view of every parameter under its ME name; keeps arrays alive."""
        keep, items = [], []
        fptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))

        def arr(t):
            a = np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))
            keep.append(a)
            return a

        for name, mod in self.named_modules():
            if isinstance(mod, _Conv):
                lp = _lib.LayerParams()
                lp.name = name.encode()
                lp.kernel = fptr(arr(mod.kernel))
                lp.K, lp.cin, lp.cout = mod.K, mod.cin, mod.cout
                if mod.bias is not None:
                    lp.bias = fptr(arr(mod.bias.reshape(-1)))
                items.append(lp)
            elif isinstance(mod, _Norm):
                lp = _lib.LayerParams()
                lp.name = name.encode()
                lp.cout = mod.bn.num_features
                lp.bn_weight, lp.bn_bias = fptr(arr(mod.bn.weight)), fptr(arr(mod.bn.bias))
                lp.bn_mean, lp.bn_var = fptr(arr(mod.bn.running_mean)), fptr(arr(mod.bn.running_var))
                items.append(lp)
        return (_lib.LayerParams * len(items))(*items), len(items), keep

    def repack(self):
        """Fold and upload the current parameters again.  ``forward`` does this on its own when ``_weights_version``
        changed; it is MANDATORY after edits that fingerprint cannot see (``p.data`` edits, writes through views)."""
        return self.pack(self._packed_device)

    def pack_host(self) -> torch.Tensor:
        """The packed blob as a CPU tensor (``eyoc_model_pack_host``: batch norms folded, fp32 fragment order + split16
        packing) - byte for byte what ``pack()`` uploads.  Needs the shared library but no GPU."""
        lib = _lib.load()
        blob = torch.zeros(self.blob_floats(), dtype=torch.float32)
        d = self._desc()
        layers, nl, keep = self._layer_params()
        _lib.check(lib.eyoc_model_pack_host(C.byref(d), layers, nl, C.c_void_p(blob.data_ptr()), blob.numel()), "eyoc_model_pack_host")
        del keep
        return blob

    def pack(self, device=None, blob: torch.Tensor | None = None, from_blob=False):
        """(Re)create the device-side model.  ``from_blob=True`` adopts an already packed blob (the
        receiving side of the weight broadcast) instead of packing this module's parameters."""
        lib = _lib.load()
        device = torch.device(device) if device is not None else self.final.kernel.device
        if device.type != "cuda":
            raise _lib.EyocError("the model runs on the GPU only: call model.to('cuda') first")
        self._invalidate()
        n = self.blob_floats()
        if blob is None:
            blob = torch.zeros(n + 64, dtype=torch.float32, device=device)
            off = ((-blob.data_ptr()) % 256) // 4
            blob = blob[off:off + n]
        d = self._desc()
        h = C.c_void_p()
        with torch.cuda.device(device):
            if from_blob:
                rc = lib.eyoc_model_create(_lib.ctx(device.index), C.byref(d), None, 0, _lib.ptr(blob), blob.numel(),
                                           C.byref(h))
            else:
                layers, nl, keep = self._layer_params()
                rc = lib.eyoc_model_create(_lib.ctx(device.index), C.byref(d), layers, nl, _lib.ptr(blob),
                                           blob.numel(), C.byref(h))
        _lib.check(rc, "eyoc_model_create")
        self._handle, self._blob, self._packed_device = h, blob, device
        self._packed_version = None if from_blob else self._weights_version()
        if self._timing:
            _lib.check(lib.eyoc_model_set_timing(h, 1), "eyoc_model_set_timing")
        if self._math != -1:
            lib.eyoc_model_set_math(h, self._math)
        if self._probe:
            _lib.check(lib.eyoc_model_set_probe(h, 1), "eyoc_model_set_probe")
        self._apply_progress()
        return blob

    @property
    def weight_blob(self):
        if self._handle is None:
            self.pack()
        return self._blob

    # ------------------------------------------------------------------ forward
    def forward(self, x: SparseTensor) -> SparseTensor:
        if not isinstance(x, SparseTensor):
            raise TypeError("expected an eyoc_amd.SparseTensor")
        if x.F.shape[1] != self.in_channels:
            raise ValueError(f"features have {x.F.shape[1]} channels, model expects {self.in_channels}")
        if self.training:
            # training mode: batch statistics + autograd (lib/trainer.py:1655-1676), layer by layer
            from .train import forward_layers
            return forward_layers(self, x)
        dev = x.device
        if self._handle is None or self._packed_device != dev:
            self.pack(dev)
        elif self._packed_version is not None and self._packed_version != self._weights_version():
            self.pack(dev)       # parameters changed in place since the blob was packed (adopted blobs are exempt)
        lib = _lib.load()
        cm = x.coordinate_manager
        maps = cm.maps(-1)        # automatic internal order (Z-order from 8192 rows); in and out stay in the caller's rows
        out = torch.empty((len(x), self.out_channels), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            ws = _lib.workspace(lib.eyoc_model_workspace_bytes(self._handle, maps), dev)

            def run():
                _lib.check(lib.eyoc_model_forward(_lib.ctx(dev.index), self._handle, maps, _lib.ptr(x.F), _lib.ptr(out),
                                                  _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "eyoc_model_forward")
            run()
            if self.range_check and lib.eyoc_model_last_math(self._handle) == 1:
                try:
                    self.check_range()
                except _lib.EyocError as e:
                    if e.code != _lib.ERR_RANGE or self._math != -1:    # a HIP error, or split16 was asked for explicitly
                        raise
                    # automatic mode: the same forward in the reference's arithmetic (fp32 MFMA has no range limit)
                    logging.warning("eyoc_amd: split16 arithmetic overflowed (an activation reached 6e4); re-running "
                                    "this forward with fp32 MFMAs - set model.spconv_math = 'fp32' for this checkpoint")
                    lib.eyoc_model_set_math(self._handle, 0)
                    try:
                        run()
                    finally:
                        lib.eyoc_model_set_math(self._handle, -1)
        return SparseTensor(out, coordinate_map_key=x.coordinate_map_key, coordinate_manager=cm)

    def check_range(self):
        """Raise ``EyocError`` (``code == EYOC_ERR_RANGE``) if ANY split16 forward since the last check stored an
        activation of magnitude >= 6e4; synchronises the current stream.  The forwards that overflowed are the ones whose
        features are NaN - forwards enqueued behind them are judged on their own (the device flag is per forward, the
        one this call reads is sticky).  Returns the largest |activation| seen since ``probe_activations(True)``
        (``None`` when the probe is off)."""
        if self._handle is None:
            return None
        mx = C.c_float(-1.0)
        with torch.cuda.device(self._packed_device):
            rc = _lib.load().eyoc_model_range_check(self._handle, _lib.stream_ptr(), C.byref(mx))
        self.last_max_activation = float(mx.value) if mx.value >= 0 else None    # valid whether or not the check raises
        _lib.check(rc, "eyoc_model_range_check")
        return self.last_max_activation

    def progress_event(self, layer: int):
        """-> a ``torch.cuda.Event`` that every forward records in front of launch ``layer`` (negative: from the end; ``None`` switches
        it off).  ``stream.wait_event(ev)`` issued AFTER a forward was enqueued waits for that forward to reach the layer."""
        if layer is None:
            self._progress = None
        else:
            ev = torch.cuda.Event()
            ev.record()                       # creates the handle
            self._progress = (ev, int(layer))
        self._apply_progress()
        return None if layer is None else self._progress[0]

    def _apply_progress(self):
        pr = getattr(self, "_progress", None)
        if self._handle is not None:
            _lib.check(_lib.load().eyoc_model_set_progress_event(self._handle, 0 if pr is None else pr[1], None if pr is None else C.c_void_p(pr[0].cuda_event)),
                       "eyoc_model_set_progress_event")

    def range_snapshot(self, words: torch.Tensor):
        """Enqueue a copy of the range guard's four device words into ``words`` (pinned ``int32[4]``) on the current stream, without
        synchronising: called right behind a forward, ``words[0] != 0`` - once an event recorded behind it has fired - says that THIS
        forward overflowed.  For callers that pipeline steps (``check_range`` reads behind everything enqueued since)."""
        assert words.is_pinned() and words.dtype == torch.int32 and words.numel() >= 4
        if self._handle is None:
            words.zero_()
            return
        with torch.cuda.device(self._packed_device):
            _lib.check(_lib.load().eyoc_model_range_snapshot(self._handle, words.data_ptr(), _lib.stream_ptr()), "eyoc_model_range_snapshot")

    def probe_activations(self, on=True):
        """Debug probe: keep the running maximum of |activation| over everything the split16 forwards store
        (``check_range()`` returns it) - tells how far a checkpoint is from the fp16 range before trusting split16."""
        self._probe = bool(on)
        if self._handle is not None:
            _lib.check(_lib.load().eyoc_model_set_probe(self._handle, 1 if on else 0), "eyoc_model_set_probe")

    # ------------------------------------------------------------------ arithmetic of the sparse convolutions
    _MATH = {"auto": -1, "fp32": 0, "split16": 1}

    @property
    def spconv_math(self) -> str:
        """"auto" (default: split16 for batches that fill the chip, else fp32), "fp32" (v_mfma_f32_16x16x4_f32) or
        "split16" (three fp16 MFMAs per product on hi/lo-split fp16 operands - 22-bit significands, fp32
        accumulation; activations must stay below 6e4: guarded - ``range_check`` / ``check_range()`` - and in "auto"
        an overflowing forward is re-run in fp32)."""
        return {v: k for k, v in self._MATH.items()}[self._math]

    @spconv_math.setter
    def spconv_math(self, mode: str):
        if mode not in self._MATH:
            raise ValueError(f"spconv_math must be one of {sorted(self._MATH)}")
        self._math = self._MATH[mode]
        if self._handle is not None:
            rc = _lib.load().eyoc_model_set_math(self._handle, self._math)
            if rc < 0:
                _lib.check(rc, "eyoc_model_set_math")

    @property
    def last_spconv_math(self) -> str:
        """What the last forward ran in ("fp32" / "split16")."""
        if self._handle is None:
            return "fp32"
        return "split16" if _lib.load().eyoc_model_last_math(self._handle) == 1 else "fp32"

    # ------------------------------------------------------------------ measurement hooks
    def set_timing(self, on: bool):
        self._timing = bool(on)
        if self._handle is not None:
            _lib.check(_lib.load().eyoc_model_set_timing(self._handle, 1 if on else 0), "eyoc_model_set_timing")

    def timing_slot(self, slot: int):
        """Select one of the two event sets the timed forwards record into / ``layer_ms`` reads from: a caller that
        alternates them can enqueue step k+1 before it reads step k's durations (no host wait between steps)."""
        _lib.check(_lib.load().eyoc_model_timing_slot(self._handle, int(slot)), "eyoc_model_timing_slot")

    def layer_ms(self):
        lib = _lib.load()
        n = lib.eyoc_model_num_layers(self._handle)
        ms = (C.c_float * n)()
        _lib.check(lib.eyoc_model_layer_ms(self._handle, ms), "eyoc_model_layer_ms")
        return list(ms)

    def layer_work(self, x: SparseTensor):
        """Per-layer algorithmic work for the geometry of ``x`` (SURVEY.md §8d formulas)."""
        lib = _lib.load()
        if self._handle is None:
            self.pack(x.device)
        n = lib.eyoc_model_num_layers(self._handle)
        names = (C.c_char_p * n)()
        pairs = (C.c_int64 * n)()
        fl, gb, cb = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
        with torch.cuda.device(x.device):
            _lib.check(lib.eyoc_model_layer_work(_lib.ctx(x.device.index), self._handle, x.coordinate_manager.maps(-1),
                                                 _lib.stream_ptr(), names, pairs, fl, gb, cb), "eyoc_model_layer_work")
        return [{"name": names[i].decode(), "pairs": int(pairs[i]), "flop": fl[i], "gather_bytes": gb[i],
                 "compulsory_bytes": cb[i]} for i in range(n)]

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.load().eyoc_model_destroy(self._handle)
        except Exception:
            pass


class ResUNetBN2(ResUNet2):
    NORM_TYPE = 'BN'


class ResUNetBN2B(ResUNet2):
    NORM_TYPE = 'BN'
    CHANNELS = [None, 32, 64, 128, 256]
    TR_CHANNELS = [None, 64, 64, 64, 64]


class ResUNetBN2C(ResUNet2):
    NORM_TYPE = 'BN'
    CHANNELS = [None, 32, 64, 128, 256]
    TR_CHANNELS = [None, 64, 64, 64, 128]


class ResUNetBN2D(ResUNet2):
    NORM_TYPE = 'BN'
    CHANNELS = [None, 32, 64, 128, 256]
    TR_CHANNELS = [None, 64, 64, 128, 128]


class ResUNetBN2E(ResUNet2):
    NORM_TYPE = 'BN'
    CHANNELS = [None, 128, 128, 128, 256]
    TR_CHANNELS = [None, 64, 128, 128, 128]


class ResUNetFatBN(ResUNet2):
    NORM_TYPE = 'BN'
    CHANNELS = [None, 32, 64, 128, 256]
    TR_CHANNELS = [None, 128, 128, 128, 256]


class ResUNetExpanded(ResUNet2):
    """model/resunet.py:254-484: ``ResUNet2`` with ``norm<i>_2`` + ``block<i>_2`` behind every stage's block.  Eval mode runs the
    packed plan like the rest of the family (``eyoc_model_desc.expanded``: the second norm of a stage is one elementwise layer, the
    second block two more staged convolutions); training mode runs layer by layer (``eyoc_amd/train.py``)."""
    EXPANDED = True
    NORM_TYPE = None


class ResUNetExpBN2C(ResUNetExpanded):
    NORM_TYPE = 'BN'
    CHANNELS = [None, 32, 64, 128, 256]
    TR_CHANNELS = [None, 64, 64, 64, 128]


MODELS = [ResUNet2, ResUNetBN2, ResUNetBN2B, ResUNetBN2C, ResUNetBN2D, ResUNetBN2E, ResUNetFatBN, ResUNetExpanded, ResUNetExpBN2C]


def load_model(name):
    """model/__init__.py:16-30 - class lookup by name; ``None`` (and a log line) if unknown."""
    mdict = {m.__name__: m for m in MODELS}
    if name not in mdict:
        logging.info(f'Invalid model index. You put {name}. Options are:')
        for m in MODELS:
            logging.info('\t* {}'.format(m.__name__))
        return None
    return mdict[name]
