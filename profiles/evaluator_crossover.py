import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from chinesechesszero_amd.net import PolicyValueNet, InferenceNet
dev = torch.device("cuda:0")
torch.manual_seed(0)
pvn = PolicyValueNet(device=dev)
inf = pvn.refresh_inference_copy()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for B in (8, 16, 32, 64, 128, 256):
    leaf = (torch.rand(B, 17, 7, 10, 9, device=dev) > 0.9).half()
    res = {}
    for fused in (1, 0):
        inf.set_options(fused_conv=bool(fused))
        with torch.no_grad(), torch.backends.cudnn.flags(enabled=True, benchmark=True):
            res[fused] = t(lambda: inf(leaf, return_logits=True))
    print(B, "fused ms %.3f" % res[1], "miopen ms %.3f" % res[0])
