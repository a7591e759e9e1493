"""Build libcczero.so in-tree with hipcc for gfx950 (used by __graft_entry__.build())."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose: bool = False) -> str:
    csrc = os.path.join(_HERE, "csrc")
    out = subprocess.run(["make", "-C", csrc], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
        print(out.stderr)
    if out.returncode != 0:
        raise RuntimeError("hipcc build of libcczero.so failed")
    path = os.path.join(_HERE, "libcczero.so")
    if not os.path.exists(path):
        raise RuntimeError("libcczero.so was not produced")
    return path
