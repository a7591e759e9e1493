#!/usr/bin/env python3
"""Dev probe: do aten::miopen_convolution_relu / _add_relu pick a fast fused kernel under MIOpen FIND mode?
(In immediate mode they fell back to the naive solver: 57 s per tower.) Hard timeouts outside."""
import sys
import time

import torch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda")
torch.backends.cudnn.benchmark = True
x = torch.randn(B, 256, 10, 9, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
z = torch.randn_like(x).contiguous(memory_format=torch.channels_last)
w = (torch.randn(256, 256, 3, 3, device=dev, dtype=torch.float16) * 0.02).contiguous(memory_format=torch.channels_last)
b = torch.randn(256, device=dev, dtype=torch.float16) * 0.1


def t(fn, name, iters=20):
    t0 = time.time()
    with torch.no_grad():
        y = fn()
    torch.cuda.synchronize()
    print(f"{name}: first call {time.time() - t0:.1f} s", flush=True)
    with torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / iters * 1e6:.1f} us", flush=True)
    return y


y0 = t(lambda: torch.nn.functional.conv2d(x, w, None, padding=1), "conv only")
y1 = t(lambda: torch.ops.aten.miopen_convolution_relu(x, w, b, [1, 1], [1, 1], [1, 1], 1), "miopen_convolution_relu")
ref1 = torch.relu(torch.nn.functional.conv2d(x, w, b, padding=1))
print("relu maxdiff", (y1 - ref1).abs().max().item(), y1.is_contiguous(memory_format=torch.channels_last), flush=True)
y2 = t(lambda: torch.ops.aten.miopen_convolution_add_relu(x, w, z, 1.0, b, [1, 1], [1, 1], [1, 1], 1), "miopen_convolution_add_relu")
ref2 = torch.relu(torch.nn.functional.conv2d(x, w, b, padding=1) + z)
print("add_relu maxdiff", (y2 - ref2).abs().max().item(), flush=True)
