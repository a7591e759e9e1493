"""The single-board path on the GPU (BASELINE configs[0]'s shape: ONE game, n_playout = 200, the way the reference's own
collect.py:133-143 calls the surface) -- VERDICT r02 item 7.

Measures, with the full 40 x 256 net:
  * sims/s of ``MCTS_AI(n_playout=200).get_action`` through the host mirror (hipGraph replay of evaluator + k_step per playout),
  * the per-playout split: graph replay wall time, the evaluator alone (eager and replayed), the simulator kernel alone,
  * the evaluator by batch size on the tower paths: k_conv3x3_small (a 16-channel x 64-pixel block per workgroup; what batches of up to 64
    boards take), the 256-pixel tile kernel (a single board = one partial tile on ONE compute unit), MIOpen + one-pass epilogue
    (what batches under 192 boards took until round 3),
so that "is MIOpen at the launch floor at B = 1?" has a number: per-layer time against the ~1.2-1.5 us kernel boundary
(MI355X_MICROARCH.md price list, row `boundary`) and the ~10-16 us graph-replay floor.

    python profiles/single_board.py > profiles/r03_single_board.json
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    from chinesechesszero_amd.game import Board
    from chinesechesszero_amd.mcts import MCTS_AI
    from chinesechesszero_amd.net import InferenceNet, PolicyValueNet
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    pvn = PolicyValueNet(device=dev)
    pvn.refresh_inference_copy()
    out = {"what": __doc__.split("\n\n")[0], "net": "random-init 40x256, fp16 inference copy (BN folded)"}

    # ---- MCTS_AI.get_action, n_playout = 200, self-play mode (tree reuse), 6 moves
    np.random.seed(0)
    player = MCTS_AI(pvn.policy_value_fn, c_puct=5, n_playout=200, is_selfplay=True)
    board = Board()
    player.get_action(board, temp=1.0)          # graph capture, MIOpen find for the batch-1 shapes
    per_move = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mv = player.get_action(board, temp=1.0)
        torch.cuda.synchronize()
        per_move.append(time.perf_counter() - t0)
        board.push(int(mv))
    med = float(np.median(per_move))
    out["mcts_ai_get_action"] = {"n_playout": 200, "seconds_per_move_median": med, "sims_per_sec": 200 / med,
                                 "moves_timed": len(per_move), "graph_replays": bool(player.mcts._graph is not None),
                                 "us_per_playout": 1e6 * med / 200}
    e = player.mcts._engine

    # ---- the pieces of one playout
    leaf = e.select_leaves()
    ev = pvn.evaluate_leaves_logits
    g = player.mcts._graph
    pieces = {}
    if g is not None:
        pieces["graph_replay_evaluator_plus_k_step_us"] = 1e6 * timed(g.replay, 200)
    pieces["evaluator_eager_us"] = 1e6 * timed(lambda: ev(leaf), 50)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        ev(leaf)
    pieces["evaluator_graph_replay_us"] = 1e6 * timed(gr.replay, 200)
    lg, v = ev(leaf)
    pieces["k_step_plus_gather_eager_us"] = 1e6 * timed(lambda: e.step_logits(lg, v), 200)
    empty = torch.cuda.CUDAGraph()
    t = torch.zeros(64, device=dev)
    with torch.cuda.graph(empty):
        t.add_(1.0)
    pieces["one_trivial_kernel_graph_replay_us"] = 1e6 * timed(empty.replay, 500)
    out["per_playout_pieces"] = pieces
    launches = 2 + 80 + 2 + 1 + 3 + 3 + 2      # pack + stem, 80 tower convolutions, heads GEMM + relu, FCs, tanh, gather, k_step
    out["per_playout_pieces"]["kernels_per_playout_approx"] = launches
    out["per_playout_pieces"]["us_per_kernel_in_the_replayed_evaluator"] = pieces["evaluator_graph_replay_us"] / launches

    # ---- the evaluator by batch size and tower path (graph-replayed: launch overhead out of the picture)
    inf = pvn._infer
    rows = []
    for B in (1, 8, 32, 64, 96, 128, 192, 1024, 4096):
        x = torch.zeros((B, 17, 7, 10, 9), dtype=torch.float16, device=dev)
        x[:, 7] = (torch.rand((B, 7, 10, 9), device=dev) > 0.9).half()
        x[:, 16] = 1.0
        for path in ("default", "small", "tile", "miopen"):
            try:
                inf.set_options(force="", fused_conv=True)
                if path in ("small", "tile"):
                    if path == "small" and B > 1024:
                        continue
                    inf.set_options(force=path)
                elif path == "miopen":
                    inf.set_options(fused_conv=False)
                with torch.backends.cudnn.flags(enabled=True, benchmark=True):
                    fn = lambda: inf(x, return_logits=True)
                    fn()
                    torch.cuda.synchronize()
                    gq = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gq):
                        fn()
                    us = 1e6 * timed(gq.replay, 30 if B <= 192 else 5)
                rows.append({"boards": B, "tower": path, "evaluator_us": us, "us_per_board": us / B, "us_per_tower_layer": us / 80})
            finally:
                inf.set_options(force="", fused_conv=True)
            print(rows[-1], file=sys.stderr, flush=True)
    out["evaluator_by_batch"] = rows
    out["reference_python_on_8_cpu_cores_sims_per_sec"] = 30.0   # the reference's own mcts.py + net.py, fp32, build container
    b1 = {r["tower"]: r for r in rows if r["boards"] == 1}
    out["reading"] = (f"batch 1: k_conv3x3_small {b1['small']['evaluator_us']:.0f} us per evaluation = {b1['small']['us_per_tower_layer']:.1f} us per tower layer; "
                      f"MIOpen path {b1['miopen']['evaluator_us']:.0f} us = {b1['miopen']['us_per_tower_layer']:.1f} us per layer (conv + epilogue = 2 launches); "
                      f"the 256-pixel tile kernel on one CU {b1['tile']['evaluator_us']:.0f} us. "
                      f"MCTS_AI at n_playout = 200: {out['mcts_ai_get_action']['sims_per_sec']:.0f} sims/s against 30 for the reference's Python on 8 cores.")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
