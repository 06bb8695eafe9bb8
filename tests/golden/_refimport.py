"""Make the reference's ``lib.trainer`` and ``scripts.test_kitti`` importable in THIS container (generator side only;
nothing here travels to the GPU box or is used by a test).

Two obstacles, neither of them arithmetic:
  * the files declare ``# -*- coding: future_fstrings -*-`` (a codec package that back-ports f-strings to Python 3.5; it
    is not installed, and f-strings are native in 3.10): the codec name is registered as an alias of utf-8;
  * module-level imports of libraries that are absent here (MinkowskiEngine, open3d, pytorch3d, easydict, tensorboardX,
    dask, nuscenes): a meta-path finder answers them with EMPTY modules - attribute access yields empty sub-modules, or
    empty classes for CamelCase names so that ``class X(ME.MinkowskiNetwork)`` parses.  No function of those libraries
    is given a body: a reference function that really calls into them fails with a TypeError and cannot be pinned this
    way.  The functions ``make_golden.py`` runs (``contrastive_hardest_negative_loss``, ``calculate_ratio_test``,
    ``get_topk_matches``, ``find_corr``, ``random_sample``, ``apply_transform``, ``evaluate_nn_dist``) are pure
    torch + numpy and run unmodified, so their outputs are the reference's outputs.
"""
import codecs
import importlib.abc
import importlib.machinery
import sys
import types

REF = "/root/reference"
ABSENT = ("MinkowskiEngine", "open3d", "pytorch3d", "easydict", "tensorboardX", "dask", "nuscenes")


def _codec(name):
    return codecs.lookup("utf-8") if name.replace("-", "_") == "future_fstrings" else None


class _Empty(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name[:1].isupper():
            obj = type(name, (), {"__init__": lambda self, *a, **k: None})
        else:
            obj = _Empty(f"{self.__name__}.{name}")
            sys.modules[obj.__name__] = obj
        setattr(self, name, obj)
        return obj


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in ABSENT:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Empty(spec.name)

    def exec_module(self, module):
        pass


_installed = False


def install():
    global _installed
    if _installed:
        return
    codecs.register(_codec)
    for name in ABSENT:                      # drop plain ModuleType placeholders an earlier import may have left
        if name in sys.modules and not isinstance(sys.modules[name], _Empty):
            del sys.modules[name]
    sys.meta_path.insert(0, _Finder())
    if REF not in sys.path:
        sys.path.insert(0, REF)
    _installed = True
