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


# what decides the kernels a bench line times and the launches around them: a committed rocprofv3 profile describes a bench run
# only while these files are the ones it was taken with (bench.py prints the hash, profiles/summarize.py stores it as ``head``)
_HASHED = ("csrc/*.h", "csrc/*.hip", "csrc/Makefile", "engine.py", "selfplay.py", "net.py", "_lib.py", "replay.py", "../include/cczero.h", "../bench.py")


def code_hash() -> str:
    """16 hex digits over the sources of the timed path (kernels, C ABI, launch loop, evaluator). There is no ``.git`` on a GPU
    box (gpurun ships a snapshot without it), so "the HEAD a profile was taken at" is this content hash, not a commit id."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for pat in _HASHED:
        for p in sorted(glob.glob(os.path.join(_HERE, pat))):
            h.update(os.path.relpath(p, _HERE).encode() + b"\0")
            with open(p, "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]
