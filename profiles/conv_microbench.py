"""Tower convolution: the fused MFMA kernel (ccz_conv3x3_c256_f16) against F.conv2d (MIOpen) + ccz_bias_act_f16.

Correctness: fp32 torch convolution of the same fp16-rounded operands on a sample of boards (first, last and a
board straddling a tile edge). Timing: HIP events around `iters` back-to-back launches of each path.
usage: python profiles/conv_microbench.py [boards] [iters]
"""
import ctypes as C
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chinesechesszero_amd import _lib  # noqa: E402

if os.environ.get("CCZ_LIB"):  # diagnostic build (libcczero_stamps.so): CONV_DBG=1,2,4,... times ablated variants
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "diag", os.environ["CCZ_LIB"])


def fused(x, w, bias32, res, y, relu=1):
    L = _lib.lib()
    _lib.check(L.ccz_conv3x3_c256_f16(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()),
                                      C.c_void_p(bias32.data_ptr()), C.c_void_p(res.data_ptr()) if res is not None else None,
                                      C.c_void_p(y.data_ptr()), x.shape[0] * 90, relu))
    return y


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1)
    cl = torch.channels_last
    x = (torch.randn(B, 256, 10, 9, generator=g) * 0.5).to(dev).half().contiguous(memory_format=cl)
    res = (torch.randn(B, 256, 10, 9, generator=g) * 0.5).to(dev).half().contiguous(memory_format=cl)
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.03).to(dev).half().contiguous(memory_format=cl)
    bias = (torch.randn(256, generator=g) * 0.1).to(dev)
    y = torch.empty_like(x)
    out = {"boards": B, "iters": iters}

    # ---- correctness on a sample of boards
    sample = sorted(set([0, 1, 2, 3, B // 2, B - 2, B - 1]) & set(range(B)))
    worst = 0.0
    for use_res in (False, True):
        fused(x, w, bias, res if use_res else None, y)
        torch.cuda.synchronize()
        ref = F.conv2d(x[sample].float(), w.float(), bias, padding=1)
        if use_res:
            ref = ref + res[sample].float()
        ref = F.relu(ref)
        err = (y[sample].float() - ref).abs().max().item()
        scale = ref.abs().max().item()
        out["max_abs_err_res%d" % use_res] = err
        out["ref_max"] = scale
        worst = max(worst, err / scale)
    # every board against the fp16 MIOpen path (loose: different rounding points)
    yref = F.relu(F.conv2d(x, w, None, padding=1).float() + bias.view(1, -1, 1, 1) + res.float())
    out["max_abs_diff_vs_miopen_all_boards"] = (y.float() - yref).abs().max().item()
    out["ok"] = bool(worst < 2e-3 and out["max_abs_diff_vs_miopen_all_boards"] < 0.05)

    # ---- timing
    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    flops = 2.0 * B * 90 * 256 * 256 * 9
    t = timeit(lambda: fused(x, w, bias, res, y))
    out["fused_res_us"] = t
    out["fused_res_tflops"] = flops / t / 1e6
    t = timeit(lambda: fused(x, w, bias, None, y))
    out["fused_nores_us"] = t
    for d in [int(v) for v in os.environ.get("CONV_DBG", "").split(",") if v]:
        out["dbg%d_nores_us" % d] = timeit(lambda: fused(x, w, bias, None, y, relu=1 | (d << 8)))
    L = _lib.lib()
    bias16 = bias.half()

    def miopen_path():
        z = F.conv2d(x, w, None, padding=1)
        _lib.check(L.ccz_bias_act_f16(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(z.data_ptr()), C.c_void_p(bias16.data_ptr()),
                                      C.c_void_p(res.data_ptr()), B * 90, 256))
        return z

    if os.environ.get("CONV_SKIP_MIOPEN", "0") != "1":
        with torch.backends.cudnn.flags(enabled=True, benchmark=True):
            t = timeit(miopen_path)
        out["miopen_plus_epilogue_us"] = t
        out["miopen_plus_epilogue_tflops"] = flops / t / 1e6
    print(json.dumps(out))
    return 0 if out["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
