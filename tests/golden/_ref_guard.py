"""Shared by the golden-vector generators (build container only): keep the read-only reference tree untouched.

The reference's ``tools.log`` writes ``<directory of tools.py>/logs/<script>.log`` -- it resolves the directory from its own
``__file__`` (tools.py:46-51), not from the working directory, so a ``chdir`` does not keep it out of ``/root/reference``. The
generators therefore (1) rebind ``log`` to a no-op in every reference module that imported it and (2) assert at the end that
nothing under the reference tree is newer than the generator's own start.
"""
import os
import sys
import time

REF = "/root/reference"
_T0 = time.time()


def silence_reference_log(*modules):
    """Rebind the reference's file logger to a no-op: in its ``tools`` module and in every module that did ``from tools import log``."""
    def _no_log(*a, **k):
        return None
    ref_tools = sys.modules.get("tools")
    if ref_tools is not None and os.path.dirname(os.path.abspath(getattr(ref_tools, "__file__", ""))) == REF:
        ref_tools.log = _no_log
    for m in modules:
        if m is not None and hasattr(m, "log"):
            m.log = _no_log


def assert_reference_untouched():
    """No file or directory under the reference tree was created or modified since this generator started."""
    newer = []
    for d, dirs, files in os.walk(REF):
        for name in dirs + files:
            p = os.path.join(d, name)
            try:
                if os.lstat(p).st_mtime >= _T0 - 1.0:
                    newer.append(p)
            except OSError:
                pass
    assert not newer, f"the generator wrote into the read-only reference tree: {newer[:5]}"
