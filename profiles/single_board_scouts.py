"""One game at a time (BASELINE configs[0]'s shape: MCTS_AI, n_playout = 200, tree reuse) with scout slots (include/cczero.h ccz_scout;
selfplay.ScoutedSearch): sims/s and evaluator calls per simulation by the number of scouts, full 40 x 256 net.

    python profiles/single_board_scouts.py > profiles/r06_single_board.json
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(pvn, scouts, n=200, moves=8):
    from chinesechesszero_amd.game import Board
    from chinesechesszero_amd.mcts import MCTS, MCTS_AI
    np.random.seed(0)
    player = MCTS_AI(pvn.policy_value_fn, c_puct=5, n_playout=n, is_selfplay=True)
    player.mcts = MCTS(pvn.policy_value_fn, 5, n, scouts=scouts)
    board = Board()
    mv = player.get_action(board, temp=1.0)          # graph capture, warm-up
    board.push(int(mv))
    per_move, played = [], []
    calls0 = player.mcts._scouted.evaluator_calls if scouts else 0
    for _ in range(moves):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mv = player.get_action(board, temp=1.0)
        torch.cuda.synchronize()
        per_move.append(time.perf_counter() - t0)
        board.push(int(mv))
        played.append(int(mv))
    med = float(np.median(per_move))
    out = {"scouts": scouts, "n_playout": n, "seconds_per_move_median": med, "sims_per_sec": n / med, "us_per_playout": 1e6 * med / n,
           "moves": played}
    if scouts:
        s = player.mcts._scouted
        out["evaluator_calls_per_simulation"] = (s.evaluator_calls - calls0) / (n * moves)
        out["rows_per_evaluator_call"] = 1 + scouts
    player.mcts._engine.check_healthy()
    if scouts == 10:   # the two pieces of a simulation, timed alone (graph replays on the engine's last position)
        s, e = player.mcts._scouted, player.mcts._engine
        s.begin_move()

        def timed(fn, iters=200):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize()
            return 1e6 * (time.perf_counter() - t0) / iters
        if s.device_loop:   # ccz_scouted_run: one launch sequence per evaluator call, hit simulations repeat on the device
            out["device_loop"] = True
            e.set_run(1, 1 << 20)
            out["pieces_us"] = {"evaluator_plus_gather_plus_one_simulation_graph_replay_plus_host_read": timed(lambda: (s._g_eval_run.replay(), e.run_outcome())),
                                "one_simulation_graph_replay_plus_host_read": (timed(lambda: (s._g_run.replay(), e.run_outcome())) if s._g_run is not None else None),
                                "host_read_alone_stream_sync": timed(lambda: e.run_outcome())}
        else:
            out["device_loop"] = False
            out["pieces_us"] = {"evaluator_plus_gather_graph_replay": timed(lambda: s._g_eval.replay()),
                                "step_scout_probe_plan_graph_replay_plus_host_read": timed(lambda: (s._g_step.replay(), e.plan_state_of_board0())),
                                "host_read_alone_stream_sync": timed(lambda: e.plan_state_of_board0())}
        e.reset_tree()
    return out


SCOUTS = tuple(int(x) for x in os.environ.get("SCOUTS", "0,3,7,10,15,31").split(","))
N_PLAYOUT = int(os.environ.get("N_PLAYOUT", "200"))   # 1600 = the reference's default PLAYOUT (parameters.py:14)
MOVES = int(os.environ.get("MOVES", "8"))


def main():
    from chinesechesszero_amd.net import PolicyValueNet
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    pvn = PolicyValueNet(device=dev)
    pvn.refresh_inference_copy()
    rows = [run(pvn, s, n=N_PLAYOUT, moves=MOVES) for s in SCOUTS]
    same = all(r["moves"] == rows[0]["moves"] for r in rows)
    print(json.dumps({"what": __doc__.split("\n\n")[0], "net": "random-init 40x256, fp16 inference copy (BN folded)",
                      "same_moves_whatever_the_scouts": same, "by_scouts": rows}, indent=1))
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
