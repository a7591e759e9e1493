"""Does splitting the evaluator batch into independent chains on several HIP streams hide the tile-count tail?
(1440 tiles on 256 CUs = 5.6 rounds paid as 6 when every convolution waits for the previous one.)
usage: python profiles/conv_streams.py [boards] [layers]
"""
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chinesechesszero_amd import _lib  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    layers = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1)
    cl = torch.channels_last
    x = torch.relu(torch.randn(B, 256, 10, 9, generator=g) * 0.5).to(dev).half().contiguous(memory_format=cl)
    y = torch.empty_like(x)
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.02).to(dev).half().contiguous(memory_format=cl)
    bias = (torch.randn(256, generator=g) * 0.1).to(dev)
    L = _lib.lib()

    def chain(stream, b0, nb):
        s = C.c_void_p(stream.cuda_stream)
        off = b0 * 90 * 256 * 2
        xp, yp = C.c_void_p(x.data_ptr() + off), C.c_void_p(y.data_ptr() + off)
        for _ in range(layers // 2):
            _lib.check(L.ccz_conv3x3_c256_f16(s, xp, C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()), None, yp, nb * 90, 1))
            _lib.check(L.ccz_conv3x3_c256_f16(s, yp, C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()), xp, xp, nb * 90, 1))

    out = {"boards": B, "layers": layers}
    for parts in (1, 2, 4, 8):
        streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
        nb = B // parts
        ts = []
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i, st in enumerate(streams):
                chain(st, i * nb, nb)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        out["parts%d_us_per_layer" % parts] = min(ts[1:]) / layers * 1e6
    print(json.dumps(out))


if __name__ == "__main__":
    main()
