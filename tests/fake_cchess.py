"""A module-shaped object with the ``cchess`` calls the reference makes, backed by THIS build's rules -- the CPU
oracle (``OracleBoard``) or the product's host view (``game.Board``, GPU) --, for running tools/probe_cchess.py against a rules
implementation whose profile is known: the probe must read back exactly the preset that was installed.

``numbering``: {build piece type 1..7 -> the piece_type this "cchess" reports}: the PIECE_TYPES numbering under test."""
from __future__ import annotations

import types

import numpy as np

LETTER = {"p": 1, "c": 2, "r": 3, "n": 4, "b": 5, "a": 6, "k": 7}   # this build's piece-type codes (include/cczero.h)


def parse_fen(fen: str):
    rows, turn, _, _, half, _ = fen.split()
    sq = np.zeros(90, np.uint8)
    for i, row in enumerate(rows.split("/")):
        rank, file = 9 - i, 0
        for ch in row:
            if ch.isdigit():
                file += int(ch)
            else:
                sq[file + 9 * rank] = LETTER[ch.lower()] + (0 if ch.isupper() else 8)
                file += 1
        assert file == 9, fen
    return sq, turn == "w", int(half)


START = "rnbakabnr/9/1c5c1/p1p1p1p1p/9/9/P1P1P1P1P/1C5C1/9/RNBAKABNR w - - 0 1"


def make_module(kind: str = "oracle", numbering=None):
    numbering = numbering or {t: t for t in range(1, 8)}

    class Move:
        def __init__(self, s):
            self.s = s

        @classmethod
        def from_uci(cls, s):
            return cls(s)

        def uci(self):
            return self.s

    class Piece:
        def __init__(self, t, color):
            self.piece_type, self.color = numbering[t], color

    class Outcome:
        def __init__(self, winner, termination):
            self.winner, self.termination = winner, termination

    class Board:
        def __init__(self, fen: str = START):
            sq, red, half = parse_fen(fen)
            if kind == "oracle":
                from oracle import OracleBoard
                self.b = OracleBoard.from_array(sq, 1 if red else 0, half)
            else:
                from chinesechesszero_amd.game import Board as HostBoard
                self.b = HostBoard(sq, red, half)

        @property
        def turn(self):
            return bool(self.b.turn)

        @property
        def halfmove_clock(self):
            return int(self.b.halfmove if kind == "oracle" else self.b.halfmove_clock)

        @property
        def legal_moves(self):
            if kind == "oracle":
                return [Move(s) for s in self.b.legal_moves]
            return [Move(m.uci()) for m in self.b.legal_moves]

        def push(self, move):
            self.b.push(move.uci())

        def piece_at(self, i):
            pc = int(self.b.squares()[i])
            return Piece(pc & 7, not bool(pc & 8)) if pc else None

        def is_game_over(self):
            return bool(self.b.is_game_over())

        def is_insufficient_material(self):
            return bool(self.b.is_insufficient_material())

        def is_fourfold_repetition(self):
            return bool(self.b.is_fourfold_repetition())

        def is_sixty_moves(self):
            return bool(self.b.is_sixty_moves())

        def outcome(self):
            o = self.b.outcome()
            return None if o is None else Outcome(o.winner, getattr(o, "termination", ""))

    return types.SimpleNamespace(Board=Board, Move=Move, RED=True, BLACK=False, __version__=f"this build's rules ({kind})", __file__=None)
