"""Cycle stamps of the tower-convolution kernel (diagnostic build libcczero_stamps.so, never the shipped library).

Per workgroup, waves 0 and 4: cycles of prologue / K loop / epilogue and the in-kernel clock (s_memtime / s_memrealtime).
usage: python profiles/conv_stamps.py [boards]
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chinesechesszero_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "build", "diag", "libcczero_stamps.so")  # make -C chinesechesszero_amd/csrc stamps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1)
    cl = torch.channels_last
    x = (torch.randn(B, 256, 10, 9, generator=g) * 0.5).to(dev).half().contiguous(memory_format=cl)
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.03).to(dev).half().contiguous(memory_format=cl)
    bias = (torch.randn(256, generator=g) * 0.1).to(dev)
    y = torch.empty_like(x)
    L = _lib.lib()
    L.ccz_debug_conv_stamps.restype = C.c_int
    L.ccz_debug_conv_stamps.argtypes = [C.c_void_p]
    out = {}
    for name, dbg in [("plain", 0)] + [("dbg%d" % int(v), int(v)) for v in os.environ.get("CONV_DBG", "").split(",") if v]:
        relu = 1 | (dbg << 8)
        for _ in range(200):  # keep the chip loaded so that the clock is the loaded clock
            _lib.check(L.ccz_conv3x3_c256_f16(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()),
                                              C.c_void_p(bias.data_ptr()), None, C.c_void_p(y.data_ptr()), B * 90, relu))
        torch.cuda.synchronize()
        st = np.zeros(2048 * 2 * 16, np.uint64)
        _lib.check(L.ccz_debug_conv_stamps(st.ctypes.data_as(C.c_void_p)))
        tiles = min(2048, (B * 90 + 255) // 256)
        st = st.reshape(2048, 2, 16)[:tiles].astype(np.int64)
        loop = st[:, :, 1] - st[:, :, 0]
        real = st[:, :, 4] - st[:, :, 3]
        rec = {"tiles_sampled": int(tiles), "loop_cycles_median": float(np.median(loop)), "loop_cycles_per_halfstep": float(np.median(loop)) / 72,
               "prologue_cycles_median": float(np.median(st[:, :, 0] - st[:, :, 10])), "epilogue_cycles_median": float(np.median(st[:, :, 2] - st[:, :, 1])),
               "clock_ghz_median": float(np.median(loop / np.maximum(real, 1)) * 0.1)}
        out[name] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
