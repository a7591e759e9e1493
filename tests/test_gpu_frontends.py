"""GPU: the 'next' rows of SURVEY 8f -- batched evaluation matches and the UCI-style loop."""
import io

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_batched_match_plays_to_completion():
    from chinesechesszero_amd.match import BatchedMatch
    from chinesechesszero_amd.net import uniform_evaluator
    from test_gpu_soak import LinearEvaluator
    B = 48
    strong = LinearEvaluator(torch.device("cuda", 0), seed=1, sharp=8.0)
    m = BatchedMatch(strong, uniform_evaluator, B, n_playout=12, seed=5, max_plies=120)
    res = m.play()
    assert res["red_wins"] + res["black_wins"] + res["draws"] == B and res["unfinished"] == 0
    assert res["plies"].min() >= 1 and res["plies"].max() <= 120
    st = m.engine.stats()
    assert st["moves"] == int(res["plies"].sum())
    # match play discards the tree after every move: the root never carries visits over
    rv = m.engine.root_children()["root_visits"]
    assert np.all((rv == 0) | (res["plies"] == 120))  # boards adjudicated at the cap keep their last search


def test_uci_loop_go_returns_a_legal_move():
    from chinesechesszero_amd.game import Board
    from chinesechesszero_amd.uci import UciLoop
    from oracle.evaluators import hash_eval

    def policy(board, red_states=None, black_states=None):
        ids = board.legal_ids()
        p, v = hash_eval(board.squares()[None, :], np.array([1 if board.turn else 0]), salt=2, scale=40.0)
        return zip(ids, p[0][ids]), np.array([[v[0]]], dtype=np.float32)

    out = io.StringIO()
    loop = UciLoop(policy_value_fn=policy, n_playout=30, out=out)
    script = ["uci", "isready", "ucinewgame", "position startpos moves b2e2 h9g7", "go nodes 40", "position startpos moves a0a5", "d", "quit"]
    for line in script:
        if not loop.handle(line):
            break
    text = out.getvalue().splitlines()
    assert "uciok" in text and "readyok" in text
    best = [l for l in text if l.startswith("bestmove")]
    assert len(best) == 1
    b = Board()
    b.push("b2e2")
    b.push("h9g7")
    assert best[0].split()[1] in [m.uci() for m in b.legal_moves]
    assert any(l.startswith("info nodes 40") for l in text)
    assert any("illegal move a0a5" in l for l in text)  # rook cannot jump its own pawn... a0a5 is blocked by a3
