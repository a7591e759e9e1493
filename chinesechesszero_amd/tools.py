"""Host mirror of reference tools.py for the rollout path: action tables, flip, softmax, decode_board, is_tie.

Same names and argument meaning as the reference (tools.py:74-272) so callers and tests read alike.
The tables come from libcczero.so (``ccz_action_table`` / ``ccz_flip_map``), i.e. from the same
compile-time table the kernels index.
"""
from __future__ import annotations

import ctypes as C
import logging

import numpy as np

from . import _lib

_log = logging.getLogger("chinesechesszero_amd")


def log(message: str, level: str = "INFO", log_path: str | None = None):
    """reference tools.py:12-71 prints with rich and appends to logs/<script>.log; stdlib logging here."""
    _log.log(getattr(logging, (level or "INFO").upper(), logging.INFO), message)


def _load_tables():
    L = _lib.lib()
    uci = C.create_string_buffer(_lib.NMOVES * 5)
    fr = (C.c_uint8 * _lib.NMOVES)()
    to = (C.c_uint8 * _lib.NMOVES)()
    _lib.check(L.ccz_action_table(uci, fr, to))
    raw = uci.raw
    names = [raw[i * 5:i * 5 + 4].decode() for i in range(_lib.NMOVES)]
    fm = (C.c_int32 * _lib.NMOVES)()
    _lib.check(L.ccz_flip_map(fm))
    return names, np.frombuffer(fr, np.uint8).copy(), np.frombuffer(to, np.uint8).copy(), np.frombuffer(fm, np.int32).copy()


def get_all_legal_moves():
    """(move_id2move_action, move_action2move_id) -- reference tools.py:172-269."""
    names, _, _, _ = _load_tables()
    return {i: s for i, s in enumerate(names)}, {s: i for i, s in enumerate(names)}


_names, MOVE_FROM, MOVE_TO, _FLIP = _load_tables()
move_id2move_action = {i: s for i, s in enumerate(_names)}
move_action2move_id = {s: i for i, s in enumerate(_names)}


# ---- rule choices that cannot be checked against the absent `cchess` module, as tables -----------
# MOVE_RANK[id]: position of move id in the iteration order of ``board.legal_moves`` (None = ascending id, this build's
# canonical order); PLANE_OF_TYPE[t]: channel of piece type t (1..7) in ``decode_board`` (default t-1, tools.py:100 under
# this build's PIECE_TYPES numbering PAWN 1 .. KING 7). ``set_rules`` changes them for the process -- the analogue of
# installing another cchess version; engines created afterwards (MCTS, BatchedSelfPlay, ...) pick them up.
MOVE_RANK: np.ndarray | None = None
PLANE_OF_TYPE: tuple = (0, 0, 1, 2, 3, 4, 5, 6)
TYPE_RANK: tuple | None = None   # major key of the order by the mover's piece type (index 1..7), None = no major key
PAWN_MOVE_RESETS_CLOCK = False   # sixty-move clock / repetition history restart on pawn moves too (python-chess `is_zeroing`)
PERPETUAL_CHECK = False          # a fourfold repetition in which one side checked throughout is lost by that side
PRESET = "canonical"             # name of the installed preset ("custom" after a set_rules call with explicit tables)


def _scan_desc_rank() -> np.ndarray:
    """move_rank of "from-square descending, then to-square descending" (the square scan of python-chess-style bitboards)."""
    order = np.lexsort((-MOVE_TO.astype(np.int64), -MOVE_FROM.astype(np.int64)))
    rank = np.empty(_lib.NMOVES, np.uint16)
    rank[order] = np.arange(_lib.NMOVES, dtype=np.uint16)
    return rank


def rule_presets() -> dict:
    """Named rule profiles for :func:`set_rules` -- every entry is a GUESS at the absent ``cchess`` module except "canonical",
    which is this build's own choice; none can be verified in this image.

    * ``"canonical"`` (default): ``legal_moves`` in ascending move id; PIECE_TYPES numbering PAWN 1, CANNON 2, ROOK 3, KNIGHT 4,
      BISHOP 5, ADVISOR 6, KING 7 (plane channel = type - 1, tools.py:100); captures alone reset the clock; no perpetual-check
      adjudication. What every golden trace without an ``order`` and the benchmark use.
    * ``"python-chess-lineage"`` [unverified recollection of python-chinese-chess as a python-chess port]: PIECE_TYPES PAWN 1,
      ROOK 2, KNIGHT 3, BISHOP 4, ADVISOR 5, KING 6, CANNON 7 (so the plane channels are P,R,N,B,A,K,C); ``legal_moves``
      iterates piece sets: non-pawn moves by from-square then to-square in DESCENDING square order, pawn moves after them;
      perpetual check loses. The profile to try FIRST with reference-trained weights: with the wrong numbering the net sees
      permuted planes."""
    return {
        "canonical": dict(),
        "python-chess-lineage": dict(move_rank=_scan_desc_rank(), type_rank=(0, 1, 0, 0, 0, 0, 0, 0),
                                     plane_of_type=(0, 0, 6, 1, 2, 3, 4, 5), perpetual_check=True),
    }


def load_preset_file(path: str, allow_unsupported: bool = False) -> dict:
    """A ``preset.json`` written by ``tools/probe_cchess.py`` (the rule choices of a REAL ``cchess`` module, probed where the
    reference runs) as the entries :func:`set_rules` takes. REFUSES (``ValueError``) a file whose probe found behaviours this
    build cannot express -- installing it would claim a parity that does not hold -- unless the caller says
    ``allow_unsupported=True`` (the differences are then logged as a WARNING)."""
    import json
    with open(path) as f:
        p = json.load(f)
    if p.get("schema") != 1:
        raise ValueError(f"{path}: not a schema-1 rule preset")
    if p.get("unsupported_differences"):
        msg = f"rule preset {path}: the probed cchess differs from this build in ways no table expresses: {p['unsupported_differences']}"
        if not allow_unsupported:
            raise ValueError(msg + " -- pass allow_unsupported=True to install the expressible part anyway")
        log(msg, "WARNING")
    mr = p.get("move_rank")
    return dict(move_rank=None if mr is None else np.asarray(mr, np.uint16), plane_of_type=tuple(p["plane_of_type"]),
                type_rank=None if p.get("type_rank") is None else tuple(p["type_rank"]),
                pawn_move_resets_clock=bool(p.get("pawn_move_resets_clock")), perpetual_check=bool(p.get("perpetual_check")))


def set_rules(move_rank=None, plane_of_type=None, type_rank=None, pawn_move_resets_clock=False, perpetual_check=False, preset=None,
              allow_unsupported: bool = False):
    """Install the rule profile of the process. Every call sets ALL choices: what is omitted returns to this build's
    default (``set_rules()`` restores them all). ``preset``: a name from :func:`rule_presets`, or the path of a ``preset.json``
    written by ``tools/probe_cchess.py`` (refused if the probe found behaviours no table expresses, unless ``allow_unsupported``);
    explicit arguments override its entries. Boards cache their legal-move list: change the profile between games."""
    global MOVE_RANK, PLANE_OF_TYPE, TYPE_RANK, PAWN_MOVE_RESETS_CLOCK, PERPETUAL_CHECK, PRESET
    explicit = any(x is not None for x in (move_rank, plane_of_type, type_rank)) or pawn_move_resets_clock or perpetual_check
    if preset is not None:
        table = rule_presets()
        if preset in table:
            p = table[preset]
        elif isinstance(preset, str) and preset.endswith(".json"):
            p = load_preset_file(preset, allow_unsupported)   # what tools/probe_cchess.py wrote after asking a real cchess module
        else:
            raise ValueError(f"unknown rule preset {preset!r}: one of {sorted(table)} or the path of a preset.json written by tools/probe_cchess.py")
        move_rank = p.get("move_rank") if move_rank is None else move_rank
        plane_of_type = p.get("plane_of_type") if plane_of_type is None else plane_of_type
        type_rank = p.get("type_rank") if type_rank is None else type_rank
        pawn_move_resets_clock = pawn_move_resets_clock or p.get("pawn_move_resets_clock", False)
        perpetual_check = perpetual_check or p.get("perpetual_check", False)
    PRESET = (preset if not explicit else "custom") if (preset is not None or explicit) else "canonical"
    PAWN_MOVE_RESETS_CLOCK = bool(pawn_move_resets_clock)
    PERPETUAL_CHECK = bool(perpetual_check)
    if type_rank is not None:
        tr = tuple(int(x) for x in type_rank)
        if len(tr) != 8 or max(tr) > 7 or min(tr) < 0:
            raise ValueError("type_rank must be 8 entries in 0..7 (index = piece type, entry 0 unused)")
        TYPE_RANK = tr if any(tr[1:]) else None
    else:
        TYPE_RANK = None
    if move_rank is not None:
        r = np.ascontiguousarray(move_rank, dtype=np.uint16)
        if r.shape != (_lib.NMOVES,) or not np.array_equal(np.sort(r), np.arange(_lib.NMOVES)):
            raise ValueError("move_rank must be a permutation of 0..2085")
        MOVE_RANK = r
    else:
        MOVE_RANK = None
    if plane_of_type is not None:
        pt = tuple(int(x) for x in plane_of_type)
        if len(pt) != 8 or sorted(pt[1:]) != list(range(7)):
            raise ValueError("plane_of_type must be 8 entries, [1..7] a permutation of 0..6")
        PLANE_OF_TYPE = (0,) + pt[1:]
    else:
        PLANE_OF_TYPE = (0, 0, 1, 2, 3, 4, 5, 6)


def current_rules() -> dict:
    """The installed profile as keyword arguments (what ``oracle.set_rules`` takes too: the checker's twin of this module)."""
    return dict(move_rank=MOVE_RANK, plane_of_type=PLANE_OF_TYPE, type_rank=TYPE_RANK,
                pawn_move_resets_clock=PAWN_MOVE_RESETS_CLOCK, perpetual_check=PERPETUAL_CHECK)


def order_ids(ids, squares=None):
    """Legal move ids in ``board.legal_moves`` order: ascending (TYPE_RANK[type of the mover], MOVE_RANK[id]); without
    installed tables that is ascending id. ``squares`` (90 piece codes) is needed only when TYPE_RANK is installed."""
    ids = list(ids)
    minor = (lambda i: i) if MOVE_RANK is None else (lambda i: int(MOVE_RANK[i]))
    if TYPE_RANK is None:
        return sorted(ids, key=minor)
    if squares is None:
        raise ValueError("a type-major order needs the position")
    return sorted(ids, key=lambda i: (TYPE_RANK[int(squares[int(MOVE_FROM[i])]) & 7], minor(i)))


def flip_map() -> np.ndarray:
    """int32[2086]: id -> id of the file-mirrored move (collect.py:118-123)."""
    return _FLIP.copy()


_FLIP_FILE = {"a": "i", "b": "h", "c": "g", "d": "f", "e": "e", "f": "d", "g": "c", "h": "b", "i": "a"}


def flip(string: str) -> str:
    """Mirror a UCI move string left-right (reference tools.py:133-166)."""
    return _FLIP_FILE[string[0]] + string[1] + _FLIP_FILE[string[2]] + string[3]


def softmax(x):
    """reference tools.py:126-129."""
    probs = np.exp(x - np.max(x))
    probs /= np.sum(probs)
    return probs


def decode_board(board):
    """Board -> (red int8[7,10,9], black int8[7,10,9]) one-hot planes (reference tools.py:74-106).

    ``board`` is anything with ``piece_at(square)`` returning an object with ``piece_type`` (1..7)
    and ``color`` (True = RED), or a :class:`chinesechesszero_amd.game.Board`.
    """
    sq = getattr(board, "squares", None)
    if callable(sq):
        s = np.asarray(sq(), dtype=np.uint8)
        red = np.zeros((7, 90), np.int8)
        black = np.zeros((7, 90), np.int8)
        occ = np.nonzero(s)[0]
        for i in occ:
            pc = int(s[i])
            (black if pc & 8 else red)[PLANE_OF_TYPE[pc & 7], i] = 1
        return red.reshape(7, 10, 9), black.reshape(7, 10, 9)
    red_state = np.zeros((7, 10, 9), dtype=np.int8)
    black_state = np.zeros((7, 10, 9), dtype=np.int8)
    for i in range(10):
        for j in range(9):
            piece = board.piece_at(j + i * 9)
            if piece:
                (red_state if piece.color else black_state)[PLANE_OF_TYPE[piece.piece_type], i, j] = 1
    return red_state, black_state


def is_tie(board) -> bool:
    """reference tools.py:109-123."""
    return board.is_insufficient_material() or board.is_fourfold_repetition() or board.is_sixty_moves()
