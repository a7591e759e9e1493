#!/usr/bin/env python3
"""k_fc_f16 (csrc/cczero_heads.h) on its own, with the ablation switches of the diagnostic build:
    make -C chinesechesszero_amd/csrc ab NAME=fcdiag ABFLAGS=-DCCZ_FC_DIAG && python profiles/fc_microbench.py
Policy shape (M rows x N 2086 x K 1536) and value shape (N 256, K 640); HIP events over 200 back-to-back launches."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chinesechesszero_amd import _lib  # noqa: E402

DIAG_DIR = "ab" if os.path.exists(os.path.join(ROOT, "build", "ab", "libcczero_ab_fcdiag.so")) else "diag"   # build/diag is not shipped to the GPU box (.gpurunignore): copy the library to build/ab for a run there
if os.path.exists(os.path.join(ROOT, "build", DIAG_DIR, "libcczero_ab_fcdiag.so")) and "--product" not in sys.argv:
    _lib.LIB_PATH = os.path.join(ROOT, "build", DIAG_DIR, "libcczero_ab_fcdiag.so")
L = _lib.lib()
dev = torch.device("cuda", 0)
s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
out = {"lib": os.path.basename(_lib.LIB_PATH), "rows": []}
for M in (3712, 4096, 2048, 1024, 128):
    for name, N, K in (("policy", 2086, 1536), ("value", 256, 640)):
        Np = -(-N // 128) * 128
        a = torch.relu(torch.randn(M, K, device=dev)).half()
        w = (torch.randn(Np, K, device=dev) * 0.03).half()
        b = torch.randn(Np, device=dev)
        c = torch.empty(M, N, dtype=torch.float16, device=dev)
        for force, kname in ((2, "k_fc_f16 128x128"), (4, "k_fc_wide_f16 256x144")):   # relu bit 1 / bit 2 force one kernel (round 4, second half)
            for _ in range(5):
                _lib.check(L.ccz_fc_f16(s, P(a), K, P(w), P(b), P(c), N, M, N, K, force, None))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                _lib.check(L.ccz_fc_f16(s, P(a), K, P(w), P(b), P(c), N, M, N, K, force, None))
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 200
            out["rows"].append({"shape": name, "M": M, "ablation": kname, "us": us, "tflops": 2.0 * M * N * K / us / 1e6})
            print(out["rows"][-1], file=sys.stderr, flush=True)
        for dbg, what in ((1, "no MFMA"), (2, "no fragment reads"), (4, "no DMA in the loop"), (8, "no epilogue"), (16, "no barrier"), (1 | 2, "DMA + barrier only"),
                          (2 | 4, "MFMA + barrier only"), (1 | 2 | 4, "barrier only")):
            if "fcdiag" not in _lib.LIB_PATH or name != "policy":
                continue
            for _ in range(5):
                _lib.check(L.ccz_fc_f16(s, P(a), K, P(w), P(b), P(c), N, M, N, K, (dbg << 8) | 4, None))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                _lib.check(L.ccz_fc_f16(s, P(a), K, P(w), P(b), P(c), N, M, N, K, (dbg << 8) | 4, None))
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 200
            out["rows"].append({"shape": name, "M": M, "ablation": "wide: " + what, "us": us, "tflops": 2.0 * M * N * K / us / 1e6})
            print(out["rows"][-1], file=sys.stderr, flush=True)
        for dbg, what in ((0, "full"), (1, "no MFMA"), (2, "no fragment reads"), (4, "no DMA in the loop"), (8, "no epilogue"), (16, "no barrier"),
                          (1 | 2, "DMA + barrier only"), (2 | 4, "MFMA + barrier only"), (2 | 4 | 16, "MFMA only")):
            if dbg and "fcdiag" not in _lib.LIB_PATH:
                continue
            for _ in range(5):
                _lib.check(L.ccz_fc_f16(s, P(a), K, P(w), P(b), P(c), N, M, N, K, (dbg << 8) | 2, None))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                _lib.check(L.ccz_fc_f16(s, P(a), K, P(w), P(b), P(c), N, M, N, K, (dbg << 8) | 2, None))
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 200
            row = {"shape": name, "M": M, "ablation": what, "us": us, "tflops": 2.0 * M * N * K / us / 1e6}
            out["rows"].append(row)
            print(row, file=sys.stderr, flush=True)
        # torch's GEMM on the same operands, for scale
        wt = w[:N].t().contiguous()
        for _ in range(5):
            torch.addmm(b[:N].half(), a, wt)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            torch.addmm(b[:N].half(), a, wt)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 200
        out["rows"].append({"shape": name, "M": M, "ablation": "torch.addmm (hipBLASLt)", "us": us, "tflops": 2.0 * M * N * K / us / 1e6})
        print(out["rows"][-1], file=sys.stderr, flush=True)
print(json.dumps(out))
