"""Batched evaluation matches: many concurrent games between two evaluators on the lockstep engine.

GPU-native form of reference ``Game.start_play`` (game.py:77-130) with two non-self-play ``MCTS_AI``
players (mcts.py:225-229): temperature 1e-3 (visit-count arg-max up to ties), no Dirichlet noise, the
tree is discarded after every move. All games start together from the opening, so every unfinished
board has the same side to move and ONE evaluator call per simulation serves the whole batch; finished
boards idle until the batch is done. (The reference's own evaluator hook is commented out,
train.py:313-319; this is SURVEY 8f row 2.)
"""
from __future__ import annotations

import numpy as np

from .engine import SelfPlayEngine
from .parameters import C_PUCT


class BatchedMatch:
    def __init__(self, evaluator_red, evaluator_black, n_boards: int, n_playout: int = 400, c_puct: float = C_PUCT,
                 seed: int = 0, device: int = 0, max_plies: int = 0, temp: float = 1e-3):
        self.ev = {1: evaluator_red, 0: evaluator_black}
        self.B = n_boards
        self.n_playout = n_playout
        self.temp = temp
        # eps = 0: the sampling distribution is pi itself (mcts.py:227), drawn from the board's Philox stream
        self.engine = SelfPlayEngine(n_boards, n_playout=n_playout, c_puct=c_puct, eps=0.0, alpha=0.2, temp=temp,
                                     seed=seed, device=device, max_plies=max_plies, mirror=False)
        self.before_move = self.on_move = None
        self._temps = np.full(n_boards, temp, np.float64)

    def play_move(self, turn: int):
        """One lockstep move of every unfinished board by the player of colour ``turn`` (1 RED, 0 BLACK): n_playout
        simulations on a fresh tree, move ~ pi at temperature 1e-3, tree discarded (mcts.py:225-229). Returns the moves
        played (host int32 [B], -1 on finished boards)."""
        e = self.engine
        ev = self.ev[turn]
        leaf = e.select_leaves()
        for i in range(self.n_playout):
            prob, value = ev(leaf)
            if i + 1 < self.n_playout:
                leaf = e.step(prob, value)
            else:
                e.expand_backup(prob, value)
        if self.before_move is not None:
            self.before_move(self)
        moves = e.finish_move(temps=self._temps, keep_tree=False).cpu().numpy()
        if self.on_move is not None:
            self.on_move(self, moves)
        return moves

    def play(self, max_moves: int = 4096, before_move=None, on_move=None):
        """Play every board to the end. Returns dict(red_wins, black_wins, draws, plies). ``before_move(match)`` runs after
        the search and before the move of every lockstep ply, ``on_move(match, moves)`` right after it (tests, logging)."""
        e = self.engine
        self.before_move, self.on_move = before_move, on_move
        self._temps = np.full(self.B, self.temp, np.float64)
        turn = 1
        for _ in range(max_moves):
            st = e.game_status()
            if st["over"].all():
                break
            self.play_move(turn)
            turn ^= 1
        st = e.game_status()
        e.check_healthy()
        w = st["winner"]
        done = st["over"] == 1
        return {"red_wins": int(((w == 1) & done).sum()), "black_wins": int(((w == 0) & done).sum()),
                "draws": int(((w == -1) & done).sum()), "unfinished": int((~done).sum()), "plies": st["plies"].copy()}
