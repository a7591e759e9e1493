import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from chinesechesszero_amd.net import PolicyValueNet
dev = torch.device("cuda:0")
torch.manual_seed(0)
pvn = PolicyValueNet(device=dev)
inf = pvn.refresh_inference_copy()
B = 4096
leaf = (torch.rand(B, 17, 7, 10, 9, device=dev) > 0.9).half()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def miopen():
    x = leaf.view(B, 119, 10, 9)
    x = torch.cat([x[:, 49:56], x[:, 105:119]], dim=1)
    x = x.to(torch.float16).contiguous(memory_format=torch.channels_last)
    return inf._epilogue(F.conv2d(x, inf.stem_w, None, padding=1), inf.stem_b)
with torch.no_grad(), torch.backends.cudnn.flags(enabled=True, benchmark=True):
    import ctypes as C
    from chinesechesszero_amd import _lib
    L = _lib.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x64 = torch.empty((B, 90, 64), dtype=torch.float16, device=dev)
    y = torch.empty((B, 256, 10, 9), dtype=torch.float16, device=dev, memory_format=torch.channels_last)
    print("pack us", t(lambda: L.ccz_pack_live_planes_f16(st, C.c_void_p(leaf.data_ptr()), C.c_void_p(x64.data_ptr()), B)))
    print("stem conv us", t(lambda: L.ccz_conv3x3_stem_f16(st, C.c_void_p(x64.data_ptr()), C.c_void_p(inf.stem_w64.data_ptr()), C.c_void_p(inf.stem_b32.data_ptr()), C.c_void_p(y.data_ptr()), B * 90, 1)))
    print("fused stem us", t(lambda: inf._stem_fused(leaf)))
    print("miopen stem us", t(miopen))
    a = inf._stem_fused(leaf); b = miopen()
    print("max diff", (a.float() - b.float()).abs().max().item())
