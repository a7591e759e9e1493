#!/usr/bin/env python3
"""Full-size soak: 4096 boards of real self-play (40x256 random-init net, fp16) played until games finish,
with harvest + restart, reporting game statistics and engine health. usage: selfplay_soak.py [boards] [sims] [moves]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chinesechesszero_amd.net import PolicyValueNet  # noqa: E402
from chinesechesszero_amd.selfplay import BatchedSelfPlay  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 48
moves = int(sys.argv[3]) if len(sys.argv) > 3 else 220
max_plies = int(sys.argv[4]) if len(sys.argv) > 4 else 200
dev = torch.device("cuda", 0)
torch.manual_seed(0)
pvn = PolicyValueNet(device=dev)
cache_log2 = int(os.environ.get("CCZ_EVAL_CACHE_LOG2", "24"))
verify = os.environ.get("CCZ_CACHE_VERIFY", "1") == "1" and cache_log2 > 0   # CCZ_FLAG_CACHE_VERIFY: ~1 hit in 128 evaluated again and compared
sp = BatchedSelfPlay(pvn.evaluate_leaves_logits, B, n_playout=n, seed=0, max_plies=max_plies, eval_cache_log2=cache_log2, cache_verify=verify)
e = sp.engine
rows = games = decisive = truncated = 0
lengths = []
t0 = time.perf_counter()
for mv in range(moves):
    sp.run_move()
    st = e.game_status()
    if st["over"].any():
        over = st["over"] == 1
        lengths += st["plies"][over].tolist()
        decisive += int((st["winner"][over] >= 0).sum())
        truncated += int(((st["plies"][over] >= max_plies) & (st["winner"][over] < 0)).sum())
        got = 0
        for s, p, z in sp.harvest_chunks(1 << 17):     # bounded device buffers whatever the burst size
            assert torch.allclose(p.sum(1), torch.ones_like(p[:, 0]), atol=1e-4)
            got += s.shape[0]
        assert got == 2 * int(st["plies"][over].sum())
        rows += got
        games += int(over.sum())
    if mv % 5 == 4:
        _s = e.stats()
        print(f"move {mv + 1}: {games} games, {rows} rows, nodes_peak {_s['nodes_peak']}, depth_peak {_s['depth_peak']}, "
              f"{time.perf_counter() - t0:.0f} s", flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = e.stats()
e.check_healthy()
out = {"boards": B, "sims_per_move": n, "moves_played_per_board": moves, "wall_s": dt, "sims_per_s": st["sims"] / dt,
       "moves_per_s": st["moves"] / dt, "games_finished": games, "decisive": decisive, "adjudicated_at_cap": truncated,
       "rows_harvested": rows, "mean_game_plies": float(np.mean(lengths)) if lengths else None,
       "min_game_plies": int(min(lengths)) if lengths else None, "error_flags": st["error_flags"],
       "nodes_peak": st["nodes_peak"], "depth_peak": st["depth_peak"], "k_bar": st["sum_children"] / max(1, st["expansions"]),
       "d_bar": st["sum_depth"] / max(1, st["sims"]), "terminal_leaf_share": st["terminal_leaves"] / max(1, st["sims"]),
       "max_plies_cap": max_plies,
       "eval_cache": {"entries_log2": cache_log2, "leaves_needing_the_net": st["cache_probes"], "hits": st["cache_hits"],
                      "served_by_another_boards_row": st["cache_shared_rows"], "stores": st["cache_stores"],
                      "fraction_skipped": (st["cache_hits"] + st["cache_shared_rows"]) / max(1, st["cache_probes"]),
                      "verify": {"hits_evaluated_again": st["cache_verified"], "mismatches": st["cache_verify_mismatches"]} if verify else None} if cache_log2 else None}
assert st["cache_verify_mismatches"] == 0
print(json.dumps(out))
