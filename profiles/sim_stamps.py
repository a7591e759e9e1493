#!/usr/bin/env python3
"""Diagnostic: where does a k_step wave spend its cycles? Uses libcczero_stamps.so (-DCCZ_STAMPS, in-kernel
s_memtime stamps; its run time is NOT quoted anywhere -- only the shares between stamps are read).

usage: python profiles/sim_stamps.py [boards] [sims_per_move] [warm_moves]
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

_lib.LIB_PATH = os.path.join(ROOT, "build", "diag", "libcczero_stamps.so")  # make -C chinesechesszero_amd/csrc stamps
from chinesechesszero_amd.selfplay import BatchedSelfPlay  # noqa: E402
from test_gpu_soak import LinearEvaluator  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda", 0)
ev = LinearEvaluator(dev, seed=0, sharp=8.0)
sp = BatchedSelfPlay(ev, B, n_playout=n, seed=0)
e = sp.engine
for _ in range(warm):
    sp.run_move()
leaf = e.select_leaves()
for _ in range(n // 2):
    p, v = ev(leaf)
    leaf = e.step(p, v)
L = _lib.lib()
L.ccz_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
names = ["kernarg + first load round (meta, leaf, root, kids, chain)", "expand: prior row + child stores", "backup: path node loads + N/Q stores", "fence", "root->LDS", "descent", "replay(make-move)", "gen A: ballots, occupancy words", "gen B: pseudo-moves",
         "gen: scan+compact", "gen C: king safety+id", "gen D: mask->ids", "repetition+status", "leaf bookkeeping", "encode+store"]
idx = [0, 8, 15, 1, 2, 3, 4, 5, 10, 11, 12, 13, 14, 6, 7, 9]
acc = np.zeros(len(names))
tot = 0.0
compact = os.environ.get("CCZ_DENSE", "0") != "1"   # default: the compact prior boundary, as bench.py uses it
for it in range(20):
    p, v = ev(leaf)
    leaf = e.step_logits(torch.log(p).contiguous(), v) if compact else e.step(p, v)
    st = np.zeros((B, 16), np.uint64)
    L.ccz_debug_stamps(e.h, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), C.c_void_p(st.ctypes.data))
    t = st[:, idx].astype(np.int64)
    d = np.diff(t, axis=1)
    ok = (d >= 0).all(axis=1) & (t[:, 0] > 0)
    acc += np.median(d[ok], axis=0)
    tot += np.median(t[ok, -1] - t[ok, 0])
acc /= 20
tot /= 20
print(json.dumps({"boards": B, "median_wave_ticks": tot, "shares": {k: round(float(x / acc.sum()), 3) for k, x in zip(names, acc)},
                  "ticks": {k: round(float(x)) for k, x in zip(names, acc)}}))
