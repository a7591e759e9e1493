"""Is the fused evaluator bit-reproducible from call to call? (B boards through stem / tower / heads twice.)
usage: python profiles/determinism_probe.py [B ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chinesechesszero_amd.net import InferenceNet, Net  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = Net(256, 6).to(dev).eval()
inf = InferenceNet(net).to(dev).eval()
for B in [int(a) for a in sys.argv[1:]] or [200, 256, 4096]:
    g = torch.Generator(device="cpu").manual_seed(B)
    x = torch.zeros((B, 17, 7, 10, 9), dtype=torch.float16)
    x[:, 7] = (torch.rand((B, 7, 10, 9), generator=g) > 0.9).half()
    x[:, 15] = (torch.rand((B, 7, 10, 9), generator=g) > 0.9).half()
    x[::2, 16] = 1
    x = x.to(dev)
    with torch.no_grad():
        s1, s2 = inf._stem_fused(x), inf._stem_fused(x)
        t1 = inf._tower_fused(s1.clone(memory_format=torch.preserve_format))
        t2 = inf._tower_fused(s2.clone(memory_format=torch.preserve_format))
        outs = [inf(x, return_logits=True) for _ in range(4)]
        import torch.nn.functional as F
        # the heads' 1x1 convolution as MIOpen runs it (what round 1 shipped) ...
        w4 = inf.head_wT.t().reshape(24, 256, 1, 1).contiguous(memory_format=torch.channels_last)
        h1 = F.relu(F.conv2d(t1, w4, inf.head_b))
        h2 = F.relu(F.conv2d(t1, w4, inf.head_b))
        # ... and as the plain GEMM on the NHWC rows that replaced it
        rows = t1.permute(0, 2, 3, 1).reshape(B * 90, 256)
        g1 = torch.addmm(inf.head_b, rows, inf.head_wT)
        g2 = torch.addmm(inf.head_b, rows, inf.head_wT)
        l1 = F.linear(g1.view(B, 90, 24)[:, :, :17].reshape(B, 1530), inf.policy_fc_w, inf.policy_fc_b)
        l2 = F.linear(g1.view(B, 90, 24)[:, :, :17].reshape(B, 1530), inf.policy_fc_w, inf.policy_fc_b)
    print(B, "stem", torch.equal(s1, s2), "tower", torch.equal(t1, t2), float((t1.float() - t2.float()).abs().max()),
          "head as MIOpen conv", torch.equal(h1, h2), "head as GEMM", torch.equal(g1, g2), "fc", torch.equal(l1, l2),
          "full logits", [torch.equal(outs[0][0], o[0]) for o in outs[1:]], "full v", [torch.equal(outs[0][1], o[1]) for o in outs[1:]],
          "tail rows differ", int(((t1 != t2).flatten(1).any(1)).sum()), flush=True)
    if not torch.equal(t1, t2):
        bad = (t1 != t2).permute(0, 2, 3, 1).reshape(B * 90, 256).any(1).nonzero().flatten()
        print("   differing pixels:", bad[:10].tolist(), "...", bad[-10:].tolist(), "count", int(bad.numel()))
