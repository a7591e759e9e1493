#!/usr/bin/env python3
"""Generate tests/golden/reference_game.{npz,json} by EXECUTING the reference's own game.py and collect.py (build container).

Pinned by running the reference's code, not a restatement of it:
  * ``Game.start_self_play`` (game.py:133-237) for ONE whole self-play game played by the reference's own ``MCTS_AI``
    (mcts.py) under ``np.random.seed``: the moves, pi of every ply, z, and the (aliased, game.py:234-237) history lists;
  * ``CollectPipeline.preprocess`` and ``flip_data`` (collect.py:64-131) applied to that game: the [17,7,10,9] float16 states
    incl. the constant turn plane (collect.py:78 reads a board that never advances), the mirrored states and pi[flip_map].
How: the modules import third-party packages that are absent here (cchess, h5py, IPython). Placeholder modules are registered
for them: ``h5py`` and ``IPython.display`` are never called on this path; ``cchess`` carries the names these modules touch
(``Board`` = the CPU oracle's duck-typed board, ``Move.from_uci / Move.uci``, ``RED``, ``BLACK``, an empty ``svg``), as
make_golden.py does for mcts.py. The evaluator is the deterministic hash evaluator of oracle/evaluators.py. So the vectors pin
the reference's game loop, temperature schedule, history bookkeeping, z assignment and tuple post-processing -- GIVEN the
oracle's rules (rules parity with cchess itself stays unpinned). Outputs are data only.
"""
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("CCZ_GOLDEN_OUT", HERE)   # where the fixtures are written (tests/test_cpu_golden_regenerates.py: a scratch directory)
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from _ref_guard import assert_reference_untouched, silence_reference_log  # noqa: E402

from oracle import OracleBoard  # noqa: E402
from oracle.evaluators import hash_eval  # noqa: E402

REF = "/root/reference"
N_PLAYOUT, SALT, SCALE, SEED = 24, 17, 40.0, 321


def load_reference():
    ph = types.ModuleType("cchess")
    ph.RED, ph.BLACK = True, False
    ph.Board = OracleBoard

    class Move:
        @staticmethod
        def from_uci(s):
            return s

        @staticmethod
        def uci(m):
            return m

    ph.Move = Move
    ph.svg = types.ModuleType("cchess.svg")
    sys.modules["cchess"] = ph
    sys.modules["cchess.svg"] = ph.svg
    sys.modules["h5py"] = types.ModuleType("h5py")            # imported by collect.py, used only by collect_data's file I/O
    ipy, disp = types.ModuleType("IPython"), types.ModuleType("IPython.display")
    disp.display = lambda *a, **k: None
    disp.SVG = lambda *a, **k: None
    ipy.display = disp
    sys.modules["IPython"], sys.modules["IPython.display"] = ipy, disp
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir("/tmp")
    import collect as ref_collect  # noqa
    import game as ref_game  # noqa
    import mcts as ref_mcts  # noqa
    os.chdir(cwd)
    silence_reference_log(ref_collect, ref_game, ref_mcts)   # tools.log writes next to tools.py whatever the working directory is (tools.py:46-51)
    return ref_game, ref_mcts, ref_collect


def main():
    ref_game, ref_mcts, ref_collect = load_reference()

    def policy(board, red_states=None, black_states=None):
        ids = board.legal_ids()
        p, v = hash_eval(board.squares()[None, :], np.array([1 if board.turn else 0]), salt=SALT, scale=SCALE)
        return zip(ids, p[0][ids]), np.array([[v[0]]], dtype=np.float32)

    player = ref_mcts.MCTS_AI(policy, c_puct=5, n_playout=N_PLAYOUT, is_selfplay=True)
    moves = []
    orig = player.get_action

    def logged(board, temp=1e-3, return_prob=False, on_playout=None):
        r = orig(board, temp=temp, return_prob=return_prob, on_playout=on_playout)
        moves.append(int(r[0] if return_prob else r))
        return r

    player.get_action = logged
    np.random.seed(SEED)
    game = ref_game.Game(OracleBoard())
    play_data = game.start_self_play(player, is_shown=False, temp=1.0, game_index=7)
    T = len(play_data)
    assert T == len(moves)
    # the reference returns the SAME (final) history lists in every tuple (game.py:234-237)
    assert all(t[0] is play_data[0][0] and t[1] is play_data[0][1] for t in play_data)
    out = {"moves": np.array(moves, dtype=np.int32),
           "pi": np.stack([np.asarray(t[2], dtype=np.float64) for t in play_data]),
           "z": np.array([t[3] for t in play_data], dtype=np.float64),
           "final_red_states": np.stack([np.asarray(s) for s in play_data[0][0]]),
           "final_black_states": np.stack([np.asarray(s) for s in play_data[0][1]]),
           "final_sq": game.board.squares(), "final_turn": np.int32(1 if game.board.turn else 0)}
    meta = {"n_playout": N_PLAYOUT, "salt": SALT, "scale": SCALE, "seed": SEED, "plies": T,
            "pi_dtype": str(np.asarray(play_data[0][2]).dtype), "z_dtype": str(np.asarray(play_data[0][3]).dtype),
            "state_dtype": str(np.asarray(play_data[0][0][0]).dtype),
            "game_over": bool(game.board.is_game_over()), "tie": bool(game.board.is_tie()),
            "winner": None if game.board.outcome() is None or game.board.outcome().winner is None else bool(game.board.outcome().winner)}

    # ---- collect.py:64-131 on that game (no file I/O: the pipeline object is made without running its __init__)
    cp = ref_collect.CollectPipeline.__new__(ref_collect.CollectPipeline)
    cp.board = OracleBoard()           # collect.py:28: a board that is never advanced
    processed = cp.preprocess(play_data)
    assert len(processed) == T and cp.episode_len == T
    st = np.stack([np.asarray(p[0]) for p in processed])
    meta["processed_state_dtype"] = str(st.dtype)
    meta["processed_state_shape"] = list(st.shape)
    assert all(np.array_equal(st[0], s) for s in st)       # quirk: every sample carries the same (final) state
    out["processed_state"] = st[0]
    out["processed_pi"] = np.stack([np.asarray(p[1]) for p in processed])
    out["processed_z"] = np.array([p[2] for p in processed], dtype=np.float64)
    flipped = cp.flip_data(processed)
    assert len(flipped) == 2 * T
    fs = np.stack([np.asarray(p[0]) for p in flipped[T:]])
    assert all(np.array_equal(fs[0], s) for s in fs)
    out["flipped_state"] = np.asarray(fs[0])
    out["flipped_pi"] = np.stack([np.asarray(p[1]) for p in flipped[T:]])
    out["flipped_z"] = np.array([p[2] for p in flipped[T:]], dtype=np.float64)
    meta["flipped_state_dtype"] = str(fs.dtype)
    # ---- Game.start_play (game.py:77-130) between two non-self-play MCTS_AI players (mcts.py:225-229: temp 1e-3, tree
    # discarded after every move): the match path as the reference's own loop runs it
    def pol(salt):
        def f(board, red_states=None, black_states=None):
            ids = board.legal_ids()
            p, v = hash_eval(board.squares()[None, :], np.array([1 if board.turn else 0]), salt=salt, scale=SCALE)
            return zip(ids, p[0][ids]), np.array([[v[0]]], dtype=np.float32)
        return f

    red = ref_mcts.MCTS_AI(pol(31), c_puct=5, n_playout=30, is_selfplay=False)
    black = ref_mcts.MCTS_AI(pol(32), c_puct=5, n_playout=30, is_selfplay=False)
    match_moves = []
    for pl in (red, black):
        o = pl.get_action

        def lg(board, temp=1e-3, return_prob=False, on_playout=None, _o=o):
            r = _o(board, temp=temp, return_prob=return_prob, on_playout=on_playout)
            match_moves.append(int(r))
            return r
        pl.get_action = lg
    np.random.seed(SEED + 1)
    g2 = ref_game.Game(OracleBoard())
    winner = g2.start_play(red, black, is_shown=False)
    out["match_moves"] = np.array(match_moves, dtype=np.int32)
    out["match_final_sq"] = g2.board.squares()
    meta["match"] = {"salts": [31, 32], "n_playout": 30, "seed": SEED + 1, "plies": len(match_moves),
                     "winner": (-1 if winner == -1 else bool(winner)), "red_player_idx": red.player, "black_player_idx": black.player}

    np.savez_compressed(os.path.join(OUT, "reference_game.npz"), **out)
    with open(os.path.join(OUT, "reference_game.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(meta)


if __name__ == "__main__":
    main()
    assert_reference_untouched()
