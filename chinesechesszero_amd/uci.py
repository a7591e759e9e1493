"""UCI-style stdin/stdout front-end on the single-board path.

The reference's README (README.md:3) advertises "standard UCI protocol" but ships no loop; what it does
use everywhere are UCI coordinate strings ("a0a1", tools.py:172-269). This is that missing loop:
``uci``, ``isready``, ``ucinewgame``, ``position startpos|fen <fen> [moves ...]``, ``go [nodes N]``,
``d``, ``quit``. The search is ``MCTS_AI`` on the HIP engine (no CPU fallback).
"""
from __future__ import annotations

import sys

import numpy as np

from .game import RED, Board, Move, _SYMBOL
from .parameters import C_PUCT, PLAYOUT

ENGINE_NAME = "cczero-mi355x"


def board_from_fen(fen: str) -> Board:
    parts = fen.split()
    if not parts:
        raise ValueError("empty FEN")
    rows = parts[0].split("/")
    if len(rows) != 10:
        raise ValueError("FEN needs 10 ranks")
    sq = np.zeros(90, np.uint8)
    for i, row in enumerate(rows):
        r, f = 9 - i, 0
        for ch in row:
            if ch.isdigit():
                f += int(ch)
            else:
                if ch.lower() not in _SYMBOL or f > 8:
                    raise ValueError(f"bad FEN row {row!r}")
                sq[f + 9 * r] = _SYMBOL[ch.lower()] + (0 if ch.isupper() else 8)
                f += 1
        if f != 9:
            raise ValueError(f"bad FEN row {row!r}")
    turn = RED if len(parts) < 2 or parts[1] in ("w", "r") else not RED
    halfmove = int(parts[4]) if len(parts) > 4 and parts[4].isdigit() else 0
    return Board(sq, turn, halfmove)


def parse_position(tokens: list[str]) -> Board:
    """tokens after 'position'."""
    if not tokens:
        raise ValueError("position needs startpos or fen")
    if tokens[0] == "startpos":
        board = Board()
        rest = tokens[1:]
    elif tokens[0] == "fen":
        end = tokens.index("moves") if "moves" in tokens else len(tokens)
        board = board_from_fen(" ".join(tokens[1:end]))
        rest = tokens[end:]
    else:
        raise ValueError("position needs startpos or fen")
    if rest and rest[0] == "moves":
        for u in rest[1:]:
            m = Move.from_uci(u)
            if m.id not in board.legal_ids():
                raise ValueError(f"illegal move {u}")
            board.push(m)
    return board


class UciLoop:
    def __init__(self, policy_value_fn=None, n_playout: int = PLAYOUT, device: int = 0, out=sys.stdout):
        self.policy_value_fn = policy_value_fn
        self.n_playout = n_playout
        self.device = device
        self.out = out
        self.board = None
        self.ai = None

    def _say(self, s: str):
        print(s, file=self.out, flush=True)

    def _player(self, nodes: int):
        from .mcts import MCTS_AI
        if self.policy_value_fn is None:
            from .net import PolicyValueNet
            self.policy_value_fn = PolicyValueNet(device=f"cuda:{self.device}").policy_value_fn
        if self.ai is None or self.ai.mcts.n_playout != nodes:
            self.ai = MCTS_AI(self.policy_value_fn, c_puct=C_PUCT, n_playout=nodes, is_selfplay=False, device=self.device)
        return self.ai

    def handle(self, line: str) -> bool:
        """Process one command line; returns False on quit."""
        tok = line.split()
        if not tok:
            return True
        cmd = tok[0]
        if cmd == "uci":
            self._say(f"id name {ENGINE_NAME}")
            self._say("id author cczero-mi355x builders")
            self._say(f"option name Playouts type spin default {self.n_playout} min 1 max 1000000")
            self._say("uciok")
        elif cmd == "isready":
            self._say("readyok")
        elif cmd == "setoption" and len(tok) >= 5 and tok[1] == "name" and tok[2].lower() == "playouts":
            self.n_playout = int(tok[4])
        elif cmd == "ucinewgame":
            self.board = Board()
            self.ai = None
        elif cmd == "position":
            try:
                self.board = parse_position(tok[1:])
            except (ValueError, KeyError) as e:
                self._say(f"info string error {e}")
        elif cmd == "go":
            if self.board is None:
                self.board = Board()
            nodes = self.n_playout
            if "nodes" in tok:
                nodes = int(tok[tok.index("nodes") + 1])
            if self.board.is_game_over():
                self._say("bestmove (none)")
                return True
            ai = self._player(nodes)
            move, probs = ai.get_action(self.board, temp=1e-3, return_prob=True)
            rc = ai.mcts.root_children()
            self._say(f"info nodes {int(rc['root_visits'])} string visits {int(rc['visits'].max())}/{int(rc['visits'].sum())}")
            self._say(f"bestmove {Move.from_id(move).uci()}")
        elif cmd == "d":
            self._say(str(self.board if self.board is not None else Board()))
        elif cmd == "quit":
            return False
        return True

    def run(self, inp=sys.stdin):
        for line in inp:
            if not self.handle(line.strip()):
                break


if __name__ == "__main__":
    UciLoop().run()
