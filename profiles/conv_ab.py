"""A/B timing of builds of the tower convolution kernel in ONE process on ONE device (interleaved rounds).

usage: python profiles/conv_ab.py libcczero.so libcczero_ab_x.so ... [--boards 4096] [--rounds 7] [--iters 10] [--res 1]
Each library is dlopen'ed separately (after torch: one shared HIP runtime); every round times `iters` back-to-back
launches of each library's ccz_conv3x3_c256_f16 with HIP events; the report is the median and minimum over rounds.
The first library is also checked against a float32 torch convolution on a sample of boards, the others against the first.
"""
import argparse
import ctypes as C
import json
import os
import statistics

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    path = os.path.join(ROOT, "chinesechesszero_amd", name)
    if not os.path.exists(path):  # A/B and diagnostic builds: build/diag (make -C chinesechesszero_amd/csrc ab NAME=...)
        path = os.path.join(ROOT, "build", "diag", name)
    L = C.CDLL(path)
    L.ccz_conv3x3_c256_f16.restype = C.c_int
    L.ccz_conv3x3_c256_f16.argtypes = [C.c_void_p] * 6 + [C.c_int64, C.c_int32]
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--boards", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--res", type=int, default=1)
    ap.add_argument("--relu-input", type=int, default=0, help="1: the input is post-ReLU (half zeros), as inside the tower")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1)
    cl = torch.channels_last
    B = a.boards
    x = (torch.randn(B, 256, 10, 9, generator=g) * 0.5).to(dev).half().contiguous(memory_format=cl)
    if a.relu_input:
        x = torch.relu(x)
    res = (torch.randn(B, 256, 10, 9, generator=g) * 0.5).to(dev).half().contiguous(memory_format=cl)
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.03).to(dev).half().contiguous(memory_format=cl)
    bias = (torch.randn(256, generator=g) * 0.1).to(dev)
    # "name.so:FLAGS" selects the relu/flags word (bit 0 ReLU, bit 1 descending tiles, bit 2 the two-workgroups-per-CU kernel)
    specs = [(n.split(":")[0], int(n.split(":")[1]) if ":" in n else 1) for n in a.libs]
    cache = {}
    libs = []
    for name, fl in specs:
        if name not in cache:
            cache[name] = load(name)
        libs.append((f"{name}:{fl}", cache[name], fl))
    ys = [torch.empty_like(x) for _ in libs]
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # flags bit 6 (bit 7 in -DCCZ_CONV4 builds): the library's group-of-16 layout (row = (g * 90 + pos) * 16 + j for board 16 g + j); the kernel gets permuted
    # copies of the same tensors and its output is permuted back before any comparison
    def to_g16(t):
        return t.permute(0, 2, 3, 1).reshape(B // 16, 16, 90, 256).permute(0, 2, 1, 3).contiguous()

    def from_g16(t):
        return t.view(B // 16, 90, 16, 256).permute(0, 2, 1, 3).reshape(B, 10, 9, 256).permute(0, 3, 1, 2)

    G16 = 64 | 1024   # bit 6: the shipped group-of-16 kernel (packed weights); bit 10: the v4 experiment (same row layout, plain weights)
    g16 = any(fl & G16 for _, _, fl in libs)
    if g16:
        xg, rg = to_g16(x), to_g16(res)
        yg = torch.empty_like(xg)
        import sys
        sys.path.insert(0, ROOT)
        from chinesechesszero_amd.net import pack_conv_weights_g16
        wp = pack_conv_weights_g16(w.permute(0, 2, 3, 1))

    def run(L, y, fl=1, name=""):
        if fl & G16:
            rc = L.ccz_conv3x3_c256_f16(s, xg.data_ptr(), (wp if (fl & 64 and "unpacked" not in name) else w).data_ptr(), bias.data_ptr(), rg.data_ptr() if a.res else None, yg.data_ptr(), B * 90, fl)
        else:
            rc = L.ccz_conv3x3_c256_f16(s, x.data_ptr(), w.data_ptr(), bias.data_ptr(), res.data_ptr() if a.res else None, y.data_ptr(), B * 90, fl)
        assert rc == 0

    for (n, L, fl), y in zip(libs, ys):
        run(L, y, fl, n)
        if fl & G16:
            y.copy_(from_g16(yg))
    torch.cuda.synchronize()
    sample = sorted({0, min(1, B - 1), B // 2, B - 1})
    ref = F.conv2d(x[sample].float(), w.float(), bias, padding=1)
    ref = F.relu(ref + res[sample].float()) if a.res else F.relu(ref)
    out = {"boards": B, "res": a.res, "err_vs_fp32": (ys[0][sample].float() - ref).abs().max().item(),
           "max_diff_vs_first": [(y.float() - ys[0].float()).abs().max().item() for y in ys]}
    ref_all = F.conv2d(x.float(), w.float(), bias, padding=1)
    ref_all = F.relu(ref_all + res.float()) if a.res else F.relu(ref_all)
    out["err_vs_fp32_all"] = [(y.float() - ref_all).abs().max().item() for y in ys]
    del ref_all
    times = {n: [] for n, _, _ in libs}
    for _ in range(a.rounds):
        for (n, L, fl), y in zip(libs, ys):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            run(L, y, fl, n)
            e0.record()
            for _ in range(a.iters):
                run(L, y, fl, n)
            e1.record()
            torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) / a.iters * 1e3)
    flops = 2.0 * B * 90 * 256 * 256 * 9
    out["us"] = {n: {"median": statistics.median(t), "min": min(t), "tflops_median": flops / statistics.median(t) / 1e6} for n, t in times.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
