"""Build ``eyoc_amd/lib/libeyoc_hip.so`` in-tree with hipcc for gfx950 (``python -m eyoc_amd.build``)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose=False, jobs=8):
    cmd = ["make", "-C", os.path.join(HERE, "csrc"), f"-j{jobs}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        sys.stdout.write(r.stdout)
        sys.stderr.write(r.stderr)
    if r.returncode != 0:
        raise RuntimeError("building libeyoc_hip.so failed")
    return os.path.join(HERE, "lib", "libeyoc_hip.so")


if __name__ == "__main__":
    print(build(verbose=True))
