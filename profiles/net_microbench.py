#!/usr/bin/env python3
"""Dev tool: time formulations of the 256-channel 3x3 residual tower at [B,256,10,9] on MI355X (PyTorch-ROCm)."""
import sys
import time

import torch
import torch.nn.functional as F

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda")
torch.manual_seed(0)


def bench(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def tower(x, ws, bs, mode):
    for i in range(0, len(ws), 2):
        if mode == "plain":
            y = F.relu_(F.conv2d(x, ws[i], bs[i], padding=1))
            y = F.conv2d(y, ws[i + 1], bs[i + 1], padding=1)
            x = F.relu_(y.add_(x))
        elif mode == "fusedop":
            y = torch.ops.aten.miopen_convolution_relu(x, ws[i], bs[i], [1, 1], [1, 1], [1, 1], 1)
            x = torch.ops.aten.miopen_convolution_add_relu(y, ws[i + 1], x, 1.0, bs[i + 1], [1, 1], [1, 1], [1, 1], 1)
        elif mode == "nobias":
            y = F.relu_(F.conv2d(x, ws[i], None, padding=1))
            y = F.conv2d(y, ws[i + 1], None, padding=1)
            x = F.relu_(y.add_(x))
        elif mode == "convonly":
            y = F.conv2d(x, ws[i], None, padding=1)
            x = F.conv2d(y, ws[i + 1], None, padding=1)
    return x


flops = 2 * B * 90 * 256 * 256 * 9 * 80
for bench_mode in (False, True):
    torch.backends.cudnn.benchmark = bench_mode
    for dtype in (torch.float16, torch.bfloat16):
        for fmt_name, fmt in (("channels_last", torch.channels_last), ("nchw", torch.contiguous_format)):
            x = torch.randn(B, 256, 10, 9, device=dev, dtype=dtype).contiguous(memory_format=fmt)
            ws = [(torch.randn(256, 256, 3, 3, device=dev, dtype=dtype) * 0.02).contiguous(memory_format=fmt) for _ in range(80)]
            bs = [torch.zeros(256, device=dev, dtype=dtype) for _ in range(80)]
            for mode in ("plain", "fusedop", "nobias", "convonly"):
                try:
                    with torch.no_grad():
                        t = bench(lambda: tower(x, ws, bs, mode))
                    print(f"benchmark={bench_mode} {str(dtype):16s} {fmt_name:13s} {mode:9s} {t*1e3:8.2f} ms  {flops/t/1e12:7.1f} TFLOP/s", flush=True)
                except Exception as e:
                    print(f"benchmark={bench_mode} {dtype} {fmt_name} {mode} FAILED {type(e).__name__}: {str(e)[:120]}", flush=True)
