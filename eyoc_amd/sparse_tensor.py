"""``SparseTensor`` with the slice of MinkowskiEngine's interface the EYOC hot path touches.

Call sites mirrored: ``ME.SparseTensor(feats.to(device), coordinates=coords.to(device))``
(scripts/test_kitti.py:143-148, util/transform_estimation.py:128-131), ``.F`` (test_kitti.py:146,150),
``ME.SparseTensor(F, coordinate_map_key=..., coordinate_manager=...)`` (model/resunet.py:187-191) and
``.decomposed_coordinates_and_features`` (lib/trainer.py:1288-1291).

The coordinate manager owns the device-side coordinate maps / rulebooks (``eyoc_maps``), built once
per coordinate set and shared by every layer of a forward pass.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class _DeviceArray:
    """A device array of the library seen through ``__cuda_array_interface__`` (what ``torch.as_tensor`` wraps without copying)."""

    def __init__(self, ptr: int, shape, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<i4", "data": (ptr, False), "version": 2}   # (torch refuses the read-only flag)
        self._owner = owner


class CoordinateManager:
    """Holds ``coords int32 [N,4]`` on the device plus the lazily built ``eyoc_maps`` handle."""

    def __init__(self, coordinates: torch.Tensor):
        if coordinates.dim() != 2 or coordinates.shape[1] != 4:
            raise ValueError(f"coordinates must be [N,4] (batch,x,y,z), got {tuple(coordinates.shape)}")
        if not coordinates.is_cuda:
            raise _lib.EyocError("SparseTensor coordinates must live on the GPU (no CPU path)")
        self.coordinates = coordinates.to(torch.int32).contiguous()
        self._maps = None
        self._ws = None
        self._maps_caller = None      # a second set in the caller's row order, built only if an accessor needs one
        self._ws_caller = None

    @property
    def device(self):
        return self.coordinates.device

    def _build(self, order: int):
        lib = _lib.load()
        n = self.coordinates.shape[0]
        with torch.cuda.device(self.device):
            ws = _lib.workspace(lib.eyoc_maps_workspace_bytes(n), self.device)
            h = C.c_void_p()
            _lib.check(lib.eyoc_maps_build_ordered(_lib.ctx(self.device.index), _lib.ptr(self.coordinates), n,
                                                   _lib.ptr(ws), ws.numel(), _lib.stream_ptr(), int(order),
                                                   C.byref(h)), "eyoc_maps_build")
        return h, ws

    def maps(self, order: int = 0):
        """The device-side maps, built on first use.  ``order``: internal row order of a build that happens now -
        ``-1`` automatic (Z-order from 8192 rows: what the network forward is fastest on - ``model(x)`` asks for it),
        ``0`` the caller's order; ``1`` Z-order.  Once built the maps stay as they are; ``row_order()`` tells which
        order that is.  The accessors below (``level_coordinates``, ``table``, ``up_order``) always answer in the
        CALLER's rows: if these maps are Z-ordered they build - once - a second set in the caller's order."""
        if self._maps is None:
            self._maps, self._ws = self._build(order)
        return self._maps

    def _caller_maps(self):
        """Maps whose internal rows are the caller's rows (what the autograd layer functions and diagnostics index
        features with): the main set when it kept the caller's order, else a second set built on first use."""
        if not _lib.load().eyoc_maps_row_order(self.maps()):
            return self._maps
        if self._maps_caller is None:
            self._maps_caller, self._ws_caller = self._build(0)
        return self._maps_caller

    def rows(self, level: int) -> int:
        return int(_lib.load().eyoc_maps_rows(self.maps(), level))

    def level_coordinates(self, level: int, internal: bool = False) -> torch.Tensor:
        """Copy of the level's coordinates ``int32 [rows,4]`` in the caller's rows (``internal=True``: in the rows of the
        maps the network forward uses - Z-ordered for large inputs, see ``row_order()``)."""
        m = self.maps() if internal else self._caller_maps()
        n = self.rows(level)
        out = torch.empty((n, 4), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().eyoc_maps_copy_coords(m, level, _lib.ptr(out), _lib.stream_ptr()), "eyoc_maps_copy_coords")
        return out

    def table(self, kind: int, level: int, internal: bool = False) -> torch.Tensor:
        """Copy of a rulebook ``int32 [27, n_out]`` in the caller's rows (the autograd layer functions index features
        with it); ``internal=True``: the forward's own (possibly Z-ordered) table."""
        m = self.maps() if internal else self._caller_maps()
        n_out = {0: self.rows(level), 1: self.rows(level + 1), 2: self.rows(level)}[kind]
        out = torch.empty((27, n_out), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().eyoc_maps_copy_table(m, kind, level, _lib.ptr(out), _lib.stream_ptr()), "eyoc_maps_copy_table")
        return out

    def table_view(self, kind: int, level: int, internal: bool = False) -> torch.Tensor:
        """The same rulebook WITHOUT a copy: an ``int32 [27, n_out]`` tensor over the library's own device array (read-only by
        contract; it keeps this manager - and with it the maps - alive).  The training forward reads its ten tables this way: ten
        allocations + device-to-device copies per iteration were 1.5 ms of an 11 ms iteration's host time."""
        m = self.maps() if internal else self._caller_maps()
        n_out = {0: self.rows(level), 1: self.rows(level + 1), 2: self.rows(level)}[kind]
        ptr = _lib.load().eyoc_maps_table(m, kind, level)
        if not ptr or n_out == 0:
            return self.table(kind, level, internal)
        return torch.as_tensor(_DeviceArray(int(ptr), (27, n_out), self), device=self.device)

    def row_order(self) -> torch.Tensor | None:
        """``int32 [rows(0)]``: caller's row of every internal row when the maps keep their rows in Z-order (built by a
        network forward on >= 8192 rows, or asked for), ``None`` when the caller's order was kept.  Level coordinates and tables are in internal rows; the
        network's input and output stay in the caller's order either way."""
        lib = _lib.load()
        if not lib.eyoc_maps_row_order(self.maps()):
            return None
        out = torch.empty((self.rows(0),), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(lib.eyoc_maps_copy_row_order(self.maps(), _lib.ptr(out), _lib.stream_ptr()), "eyoc_maps_copy_row_order")
        return out

    def up_order(self, level: int, internal: bool = False) -> torch.Tensor:
        """Copy of the row order ``int32 [rows(level)]`` the transposed convolutions tile their outputs in."""
        m = self.maps() if internal else self._caller_maps()
        out = torch.empty((self.rows(level),), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().eyoc_maps_copy_up_order(m, level, _lib.ptr(out), _lib.stream_ptr()), "eyoc_maps_copy_up_order")
        return out

    def info(self, conv1_kernel_size: int = 0) -> dict:
        info = _lib.MapsInfo()
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().eyoc_maps_info(_lib.ctx(self.device.index), self.maps(), conv1_kernel_size,
                                                  _lib.stream_ptr(), C.byref(info)), "eyoc_maps_info")
        L = info.n_levels
        return {"rows": list(info.rows[:L]), "pairs_s1": list(info.pairs_s1[:L]),
                "pairs_down": list(info.pairs_down[:L - 1]), "pairs_up": list(info.pairs_up[:L - 1]),
                "pairs_conv1": int(info.pairs_conv1)}

    def __del__(self):
        try:
            for name in ("_maps", "_maps_caller"):
                if getattr(self, name, None) is not None:
                    _lib.load().eyoc_maps_free(getattr(self, name))
                    setattr(self, name, None)
        except Exception:
            pass


class SparseTensor:
    def __init__(self, features: torch.Tensor, coordinates: torch.Tensor | None = None, device=None,
                 coordinate_map_key=None, coordinate_manager: CoordinateManager | None = None, **_unused):
        if device is not None:
            features = features.to(device)
            if coordinates is not None:
                coordinates = coordinates.to(device)
        if coordinate_manager is None:
            if coordinates is None:
                raise ValueError("either coordinates or coordinate_manager is required")
            coordinate_manager = CoordinateManager(coordinates)
        if features.dim() != 2 or features.shape[0] != coordinate_manager.coordinates.shape[0]:
            raise ValueError("features must be [N,C] with one row per coordinate")
        if not features.is_cuda:
            raise _lib.EyocError("SparseTensor features must live on the GPU (no CPU path)")
        self._F = features.to(torch.float32).contiguous()
        self.coordinate_manager = coordinate_manager
        self.coordinate_map_key = coordinate_map_key if coordinate_map_key is not None else (1, 1, 1)

    @property
    def F(self):
        return self._F

    @property
    def C(self):
        return self.coordinate_manager.coordinates

    @property
    def device(self):
        return self._F.device

    def __len__(self):
        return self._F.shape[0]

    @property
    def decomposed_coordinates_and_features(self):
        """Per-batch-index lists ``(coords int32 [n_b,3], feats [n_b,C])`` (lib/trainer.py:1288-1291)."""
        b = self.C[:, 0]
        coords, feats = [], []
        for i in range(int(b.max().item()) + 1 if len(b) else 0):
            m = b == i
            coords.append(self.C[m][:, 1:])
            feats.append(self._F[m])
        return coords, feats

    def __repr__(self):
        return f"SparseTensor(N={len(self)}, C={self._F.shape[1]}, device={self.device})"
