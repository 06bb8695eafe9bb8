"""CPU oracle for the EYOC registration hot path.  TEST INFRASTRUCTURE - NOT A PRODUCT PATH.

Plain numpy / CPU-torch restatements of the algorithms the reference runs for this path. Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this
package, and only as the checker.  ``eyoc_amd`` never imports it: the product path calls the HIP
library through the C ABI and raises if that library is missing.

Pinning status:
  * ``matching.py``, ``pose.py``, ``sc2pcr.py``: pinned against golden vectors produced by
    importing the reference's own pure-torch functions (``tests/golden/make_golden.py``).
  * ``coords.py`` + ``resunet.py`` (MinkowskiEngine semantics) and ``ransac.py`` (Open3D semantics):
    **parity unpinned** - those libraries are un-vendored, un-pinned dependencies that cannot be
    installed here, and the reference holds no test or golden file for them.  They restate the
    published MinkowskiEngine 0.5.x / Open3D >= 0.12 algorithms and are pinned only by
    self-consistency checks (dense ``conv3d`` equivalence, hand-computed toy cases).
  * ``voxelize.py`` (``ME.utils.sparse_quantize``) and ``labels.py`` (``lib/trainer.py:993-1218``, which imports
    MinkowskiEngine / pytorch3d / open3d at module level): **parity unpinned** for the same reason; restated from
    the source lines cited in each function, checked against independent numpy formulations.
"""
