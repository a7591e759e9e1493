"""Host mirror of reference game.py for the rollout path: ``Board`` (the ``cchess.Board`` calls the
reference makes) and ``Game.start_self_play`` / ``start_play`` (game.py:77-237).

``Board`` is a host-side VIEW: it stores squares, side to move, the half-move clock and the move
stack, but every rule it answers (legal moves, check, insufficient material) is computed by the HIP
engine's stateless movegen kernel (``ccz_legal_moves``); there is no CPU rules implementation in the
product. Repetition is bookkeeping over the stored positions.
"""
from __future__ import annotations

import numpy as np

from . import tools
from .tools import MOVE_FROM, MOVE_TO, decode_board, is_tie, log, move_action2move_id, move_id2move_action

RED = True
BLACK = False
PAWN, CANNON, ROOK, KNIGHT, BISHOP, ADVISOR, KING = range(1, 8)  # PIECE_TYPES numbering assumed for cchess

_START_ROWS = ["RNBAKABNR", ".........", ".C.....C.", "P.P.P.P.P", ".........",
               ".........", "p.p.p.p.p", ".c.....c.", ".........", "rnbakabnr"]
_SYMBOL = {"p": PAWN, "c": CANNON, "r": ROOK, "n": KNIGHT, "b": BISHOP, "a": ADVISOR, "k": KING}
_SYMBOL_INV = {v: k for k, v in _SYMBOL.items()}


def start_squares() -> np.ndarray:
    sq = np.zeros(90, np.uint8)
    for r, row in enumerate(_START_ROWS):
        for f, ch in enumerate(row):
            if ch != ".":
                sq[f + 9 * r] = _SYMBOL[ch.lower()] + (0 if ch.isupper() else 8)
    return sq


class Move:
    """``cchess.Move`` stand-in: a UCI coordinate string ("a0a1") with its action id."""

    __slots__ = ("_uci", "id")

    def __init__(self, uci: str):
        self._uci = uci
        self.id = move_action2move_id[uci]

    @classmethod
    def from_uci(cls, uci: str) -> "Move":
        return cls(uci)

    @classmethod
    def from_id(cls, mid: int) -> "Move":
        return cls(move_id2move_action[int(mid)])

    def uci(self) -> str:
        return self._uci

    def __str__(self):
        return self._uci

    def __repr__(self):
        return f"Move.from_uci({self._uci!r})"

    def __eq__(self, other):
        return isinstance(other, Move) and other._uci == self._uci

    def __hash__(self):
        return hash(self._uci)


class Piece:
    __slots__ = ("piece_type", "color")

    def __init__(self, piece_type: int, color: bool):
        self.piece_type = piece_type
        self.color = color

    def symbol(self) -> str:
        s = _SYMBOL_INV[self.piece_type]
        return s.upper() if self.color else s


class Outcome:
    __slots__ = ("winner", "termination")

    def __init__(self, winner, termination):
        self.winner = winner
        self.termination = termination


def _move_id(move) -> int:
    if isinstance(move, Move):
        return move.id
    if isinstance(move, str):
        return move_action2move_id[move]
    return int(move)


class Board:
    def __init__(self, squares=None, turn: bool = RED, halfmove: int = 0, device: int = 0):
        self._sq = start_squares() if squares is None else np.array(squares, dtype=np.uint8).reshape(90).copy()
        self.turn = bool(turn)
        self.halfmove_clock = int(halfmove)
        self.move_stack: list[Move] = []
        self._chain = [(self._sq.tobytes(), self.turn)]  # positions since the last capture, current one last
        self._device = device
        self._cache = None
        self._start = (self._sq.copy(), self.turn, self.halfmove_clock)

    # ---- plain state ---------------------------------------------------------------------------
    def squares(self) -> np.ndarray:
        return self._sq.copy()

    def copy(self) -> "Board":
        b = Board.__new__(Board)
        b._sq = self._sq.copy()
        b.turn = self.turn
        b.halfmove_clock = self.halfmove_clock
        b.move_stack = list(self.move_stack)
        b._chain = list(self._chain)
        b._device = self._device
        b._cache = self._cache
        b._start = self._start
        return b

    def piece_at(self, square: int):
        pc = int(self._sq[square])
        return Piece(pc & 7, not bool(pc & 8)) if pc else None

    def peek(self):
        return self.move_stack[-1] if self.move_stack else None

    def push(self, move) -> None:
        """Make a move (assumed legal, as in cchess); a capture resets the half-move clock."""
        mid = _move_id(move)
        fr, to = int(MOVE_FROM[mid]), int(MOVE_TO[mid])
        capture = self._sq[to] != 0 or (tools.PAWN_MOVE_RESETS_CLOCK and (int(self._sq[fr]) & 7) == PAWN)  # "zeroing" move
        self._sq[to] = self._sq[fr]
        self._sq[fr] = 0
        self.turn = not self.turn
        self.move_stack.append(move if isinstance(move, Move) else Move.from_id(mid))
        if capture:
            self.halfmove_clock = 0
            self._chain = []
        else:
            self.halfmove_clock += 1
        self._chain.append((self._sq.tobytes(), self.turn))
        self._cache = None

    def fen(self) -> str:
        rows = []
        for r in range(9, -1, -1):
            row, empty = "", 0
            for f in range(9):
                pc = int(self._sq[f + 9 * r])
                if not pc:
                    empty += 1
                    continue
                if empty:
                    row += str(empty)
                    empty = 0
                s = _SYMBOL_INV[pc & 7]
                row += s if pc & 8 else s.upper()
            rows.append(row + (str(empty) if empty else ""))
        return "/".join(rows) + (" w" if self.turn else " b") + f" - - {self.halfmove_clock} {len(self.move_stack) // 2 + 1}"

    def __str__(self):
        return "\n".join("".join((_SYMBOL_INV[int(p) & 7].upper() if not int(p) & 8 else _SYMBOL_INV[int(p) & 7]) if p else "."
                                 for p in self._sq[9 * r:9 * r + 9]) for r in range(9, -1, -1))

    # ---- rules: answered by the GPU ------------------------------------------------------------
    def _rules(self):
        if self._cache is None:
            from .engine import legal_moves
            mask, cnt, flags = legal_moves(self._sq[None, :], np.array([1 if self.turn else 0], np.uint8),
                                           np.array([self.halfmove_clock], np.int32), device=self._device)
            self._cache = (tools.order_ids(np.nonzero(mask[0])[0].astype(np.int64).tolist(), self._sq), int(flags[0]))
        return self._cache

    def legal_ids(self) -> list[int]:
        """Legal move ids in ``legal_moves`` order: ascending id (the canonical order of this build) or ascending
        ``tools.MOVE_RANK`` when :func:`tools.set_rules` installed another order."""
        return list(self._rules()[0])

    @property
    def legal_moves(self):
        return [Move.from_id(i) for i in self._rules()[0]]

    def is_check(self) -> bool:
        return bool(self._rules()[1] & 1)

    def is_checkmate(self) -> bool:
        return self.is_check() and not self._rules()[0]

    def is_stalemate(self) -> bool:
        return not self.is_check() and not self._rules()[0]

    def is_insufficient_material(self) -> bool:
        return bool(self._rules()[1] & 2)

    def is_sixty_moves(self) -> bool:
        return bool(self._rules()[1] & 4)

    def is_fourfold_repetition(self) -> bool:
        cur = self._chain[-1]
        return sum(1 for p in self._chain if p == cur) >= 4

    def is_game_over(self) -> bool:
        return (not self._rules()[0]) or is_tie(self)

    def outcome(self):
        if not self._rules()[0]:  # mate and stalemate both lose for the side to move
            return Outcome(not self.turn, "checkmate" if self.is_check() else "stalemate")
        if self.is_insufficient_material():
            return Outcome(None, "insufficient_material")
        if self.is_sixty_moves():
            return Outcome(None, "sixty_moves")
        if self.is_fourfold_repetition():
            w = self.perpetual_check_winner() if tools.PERPETUAL_CHECK else None
            return Outcome(w, "fourfold_repetition" if w is None else "perpetual_check")
        return None

    def perpetual_check_winner(self):
        """CCZ_RULE_PERPETUAL_CHECK on the host view: in a fourfold repetition, inside the window of
        positions after the EARLIEST occurrence of the repeated position, a side whose every move gave check while not every
        move of the other side did loses. Returns the winner (RED / BLACK) or None. The check flags of the window's positions
        come from ONE batched ``ccz_legal_moves`` call."""
        cur = self._chain[-1]
        if sum(1 for p in self._chain if p == cur) < 4:
            return None
        first = next(i for i, p in enumerate(self._chain) if p == cur)
        window = self._chain[first + 1:]
        from .engine import legal_moves
        sq = np.stack([np.frombuffer(p[0], np.uint8) for p in window])
        _, _, flags = legal_moves(sq, np.array([1 if p[1] else 0 for p in window], np.uint8), None, device=self._device)
        checked = [bool(f & 1) for f in flags]
        mover = [c for p, c in zip(window, checked) if p[1] == self.turn]        # positions the side that just moved created
        other = [c for p, c in zip(window, checked) if p[1] != self.turn]
        mover_all, other_all = all(mover), bool(other) and all(other)
        if mover_all and not other_all:
            return self.turn            # the side that kept checking (it just moved) loses
        if other_all and not mover_all:
            return not self.turn
        return None


class Game:
    """reference game.py:11-237 (self-play and match loops; the SVG viewer hook is out of scope)."""

    def __init__(self, board=None, reference_quirks: bool = False, progress=None, viewer=None):
        """``progress(game_index, step, advanced, total, avg_step_seconds)``: the sink the reference feeds its rich progress
        bar from (game.py:162-185), called through ``get_action``'s ``on_playout``. ``viewer``: anything with
        ``update_board(board_text_or_svg, status_text)`` -- the hook ``graphic`` drives (reference frontend.py ``ChessWindow``; here e.g. ``examples/viewer.py``,
        game.py:47-75); without one ``graphic`` logs the text board."""
        self.board = board if board is not None else Board()
        self.reference_quirks = reference_quirks
        self.progress = progress
        self.viewer = viewer
        self.red_states = None
        self.black_states = None
        self.reset_states_history()

    def reset_states_history(self):
        """game.py:23-34"""
        init_red_state, init_black_state = decode_board(self.board)
        self.red_states = [init_red_state.copy() for _ in range(8)]
        self.black_states = [init_black_state.copy() for _ in range(8)]

    def update_states_history(self):
        """game.py:36-44: newest first"""
        red_state, black_state = decode_board(self.board)
        self.red_states.pop()
        self.red_states.insert(0, red_state)
        self.black_states.pop()
        self.black_states.insert(0, black_state)

    def graphic(self, board):
        """game.py:47-75: push the position to the viewer window (status line as the reference formats it); the SVG
        rendering of ``cchess.svg`` is not part of this build: :func:`boardsvg.board_svg` draws the position."""
        current_player = "red" if board.turn == RED else "black"
        status_text = f"to move: {current_player} - ply: {len(board.move_stack)}"
        if self.viewer is not None:
            try:
                from .boardsvg import board_svg
                last = board.peek()
                lm = (int(MOVE_FROM[last.id]), int(MOVE_TO[last.id])) if last is not None else None
                self.viewer.update_board(board_svg(board.squares(), lm), status_text)
                return
            except Exception as e:  # the reference falls back to terminal display when the window fails
                log(f"viewer update failed: {e}", "WARNING")
        log(status_text + "\n" + str(board))

    def start_play(self, player1, player0, is_shown=False):
        """game.py:77-130: player1 = RED moves first; returns the winner (True/False) or -1 for a draw."""
        self.board = Board()
        player1.set_player_idx(1)
        player0.set_player_idx(0)
        players = {RED: player1, BLACK: player0}
        while True:
            move = players[self.board.turn].get_action(self.board)
            self.update_states_history()
            self.board.push(move if isinstance(move, (Move, str)) else Move.from_id(move))
            if is_shown:
                self.graphic(self.board)
            if self.board.is_game_over():
                outcome = self.board.outcome()
                return outcome.winner if (outcome is not None and outcome.winner is not None) else -1

    def start_self_play(self, player, is_shown=False, temp=1.0, game_index=None):
        """game.py:133-237: returns [(red_states[8], black_states[8], pi[2086], z)] for one game.

        By default every sample carries its OWN 8-ply history; ``reference_quirks=True`` reproduces the
        reference's aliasing of the final history lists into every sample (game.py:234-237).
        """
        self.board = Board()
        self.reset_states_history()
        import time
        mcts_probs, current_players, histories = [], [], []
        move_count = 0
        avg_step_time = 0.0
        total = getattr(getattr(player, "mcts", None), "n_playout", 0)  # game.py:173 reads player.mcts.n_playout
        while True:
            move_count += 1
            current_temp = temp if move_count <= 30 else max(0.1, temp * 0.5)  # game.py:159
            step_start_time = time.time()
            on_playout = None
            if self.progress is not None:  # game.py:162-185: playout progress and the average time per step
                on_playout = lambda d, _s=move_count, _a=avg_step_time: self.progress(game_index, _s, d, total, _a)
            move, move_probs = player.get_action(self.board, temp=current_temp, return_prob=True, on_playout=on_playout)
            step_time = time.time() - step_start_time
            avg_step_time = (avg_step_time * (move_count - 1) + step_time) / move_count
            prob_sum = np.sum(move_probs)
            if prob_sum > 0:
                move_probs = move_probs / prob_sum  # game.py:188-190
            else:
                raise RuntimeError(f"move_probs are all zero at step {move_count}")  # reference would spin (game.py:191-193)
            mcts_probs.append(move_probs)
            current_players.append(self.board.turn)
            self.update_states_history()
            histories.append(([s for s in self.red_states], [s for s in self.black_states]))
            self.board.push(Move.from_id(move))
            if is_shown:
                self.graphic(self.board)
            if self.board.is_game_over() or is_tie(self.board):
                outcome = self.board.outcome() if self.board.is_game_over() else None
                winner_z = np.zeros(len(current_players))
                if outcome and outcome.winner is not None:
                    for i, player_id in enumerate(current_players):
                        winner_z[i] = 1 if player_id == outcome.winner else -1
                player.reset_player()
                if self.reference_quirks:
                    return [(self.red_states, self.black_states, mcts_probs[i], winner_z[i]) for i in range(len(mcts_probs))]
                return [(histories[i][0], histories[i][1], mcts_probs[i], winner_z[i]) for i in range(len(mcts_probs))]
