#!/usr/bin/env python3
"""Dev tool: isolate the simulator kernels on MI355X (no net): per-kernel HIP-event timings on developed trees.

usage: python profiles/sim_microbench.py [boards] [sims_per_move] [warm_moves]
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from chinesechesszero_amd import _lib  # noqa: E402

if os.environ.get("CCZ_LIB"):  # A/B: load another build of the library
    _lib.LIB_PATH = os.path.join(ROOT, "build", "diag", os.environ["CCZ_LIB"])
from chinesechesszero_amd.selfplay import BatchedSelfPlay  # noqa: E402
from test_gpu_soak import LinearEvaluator  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
warm_moves = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda", 0)
ev = LinearEvaluator(dev, seed=0, sharp=8.0)
sp = BatchedSelfPlay(ev, B, n_playout=n, seed=0)
e = sp.engine
for _ in range(warm_moves):
    sp.run_move()
# develop the current move's tree half way
leaf = e.select_leaves()
for _ in range(n // 2):
    p, v = ev(leaf)
    leaf = e.step(p, v)
p, v = ev(leaf)
torch.cuda.synchronize()
s0 = e.stats()


def timed(fn, iters):
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in evs]) * 1e3
    return float(np.median(t)), float(t.mean())


out = {"boards": B, "sims_per_move": n, "warm_moves": warm_moves}
# fused step with a FIXED evaluator output (same p, v every time): trees keep growing realistically
out["k_step_us(median,mean)"] = timed(lambda: e.step(p, v), 100)
s1 = e.stats()
out["k_bar"] = (s1["sum_children"] - s0["sum_children"]) / max(1, s1["expansions"] - s0["expansions"])
out["d_bar"] = (s1["sum_depth"] - s0["sum_depth"]) / max(1, s1["sims"] - s0["sims"])
out["k_select_us"] = timed(lambda: e.select_leaves(), 50)          # repeated select of the same leaf (no mutation)
def both():
    e.expand_backup(p, v)
    e.select_leaves()
out["expand+select_2launch_us"] = timed(both, 50)
# stateless movegen on the current root positions
sq = torch.from_numpy(np.pad(e.root_positions(), ((0, 0), (0, 6)))).to(dev)
turn = torch.from_numpy(e.game_status()["turn"]).to(dev)
mask = torch.zeros((B, 66), dtype=torch.int32, device=dev)
cnt = torch.zeros(B, dtype=torch.int32, device=dev)
L = _lib.lib()
stream = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
ptr = lambda t: C.c_void_p(t.data_ptr())
out["k_legal_moves_us"] = timed(lambda: L.ccz_legal_moves(stream(), B, ptr(sq), ptr(turn), None, ptr(mask), ptr(cnt), None), 50)
x = torch.zeros(1 << 20, device=dev)
out["tiny_torch_kernel_us"] = timed(lambda: x.add_(1.0), 50)
e.check_healthy()
st = e.stats()
out["depth_peak"] = st["depth_peak"]
out["nodes_peak"] = st["nodes_peak"]
print(json.dumps(out))
