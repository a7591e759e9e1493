#!/usr/bin/env python3
"""Measurement for SURVEY 8f row 2 (evaluation matches on the batched engine) and row 1 (tuple sink):
B concurrent games between two random-init nets, n sims/move, played to the end (cap), then games/s, moves/s,
sims/s; and the rows/s of harvest -> TupleSink (.npy trainer format). usage: match_bench.py [boards] [sims] [cap]"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chinesechesszero_amd.collect import TupleSink  # noqa: E402
from chinesechesszero_amd.match import BatchedMatch  # noqa: E402
from chinesechesszero_amd.net import PolicyValueNet  # noqa: E402
from chinesechesszero_amd.selfplay import BatchedSelfPlay  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 120
blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 40
dev = torch.device("cuda", 0)
torch.manual_seed(0)
a = PolicyValueNet(device=dev, resblocks_num=blocks)
torch.manual_seed(1)
b = PolicyValueNet(device=dev, resblocks_num=blocks)
m = BatchedMatch(a.evaluate_leaves, b.evaluate_leaves, B, n_playout=n, seed=0, max_plies=cap)
a.evaluate_leaves(m.engine.leaf_input)
b.evaluate_leaves(m.engine.leaf_input)  # MIOpen find outside the timed region
torch.cuda.synchronize()
t0 = time.perf_counter()
res = m.play()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = m.engine.stats()
out = {"match": {"boards": B, "sims_per_move": n, "ply_cap": cap, "net": f"{blocks}x256 fp16 (two random-init nets)", "wall_s": dt,
                 "games_per_s": B / dt, "moves_per_s": st["moves"] / dt, "sims_per_s": st["sims"] / dt,
                 "red_wins": res["red_wins"], "black_wins": res["black_wins"], "draws": res["draws"],
                 "mean_plies": float(res["plies"].mean())}}
# tuple sink: self-play with a short cap so that every board finishes, harvest rows -> .npy files
sp = BatchedSelfPlay(a.evaluate_leaves_logits, B, n_playout=8, seed=0, max_plies=24)
rows = 0
t_harvest = t_sink = 0.0
with tempfile.TemporaryDirectory() as d:
    sink = TupleSink(d)
    for _ in range(26):
        sp.run_move()
        stt = sp.engine.game_status()
        if stt["over"].any():
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for s, p, z in sp.harvest_chunks(1 << 17):
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                sink.append(s, p, z, games=0)
                t_harvest += t2 - t1
                t_sink += time.perf_counter() - t2
                rows += s.shape[0]
                t1 = time.perf_counter()
    t3 = time.perf_counter()
    n_written = sink.flush()
    t_sink += time.perf_counter() - t3
    size = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
out["tuple_sink"] = {"rows": rows, "rows_written": n_written, "harvest_rows_per_s": rows / max(t_harvest, 1e-9),
                     "sink_rows_per_s": rows / max(t_sink, 1e-9), "bytes_on_disk": size,
                     "harvest_GBps": rows * (21420 + 8344 + 4) / max(t_harvest, 1e-9) / 1e9}
print(json.dumps(out))
