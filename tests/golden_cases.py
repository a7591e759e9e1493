"""Start positions of the golden search traces (data mirrored from tests/golden/make_golden.py)."""
import numpy as np


def sq(name: str) -> int:
    return (ord(name[0]) - 97) + 9 * int(name[1])


def _place(spec):
    b = np.zeros(90, dtype=np.uint8)
    for name, pc in spec.items():
        b[sq(name)] = pc
    return b


STARTS = {
    "two_rooks": _place({"d0": 7, "a7": 3, "b8": 3, "e9": 15}),
    "capture_to_bare": _place({"e0": 7, "d0": 6, "e1": 9, "d9": 15, "c9": 13}),
    "rook_knight": _place({"e0": 7, "e1": 6, "c2": 4, "h4": 3, "d9": 15, "e8": 14, "a5": 11, "g6": 9}),
    "wide80": _place({"e1": 7, "a2": 3, "i7": 3, "b4": 2, "h5": 2, "c3": 4, "g6": 4, "a6": 1, "c7": 1, "e6": 1, "g7": 1, "i6": 1,
                      "d0": 6, "f0": 6, "c0": 5, "g0": 5, "d9": 15, "e8": 14, "a9": 11}),
    "pawns": _place({"d0": 7, "a3": 1, "c3": 1, "e3": 1, "g3": 1, "i3": 1, "f9": 15, "a6": 9, "c6": 9, "e6": 9, "g6": 9, "i6": 9}),
}

START_ROWS = ["RNBAKABNR", ".........", ".C.....C.", "P.P.P.P.P", ".........",
              ".........", "p.p.p.p.p", ".c.....c.", ".........", "rnbakabnr"]
_PC = {"p": 1, "c": 2, "r": 3, "n": 4, "b": 5, "a": 6, "k": 7}


def start_position() -> np.ndarray:
    b = np.zeros(90, dtype=np.uint8)
    for r, row in enumerate(START_ROWS):
        for f, ch in enumerate(row):
            if ch != ".":
                b[f + 9 * r] = _PC[ch.lower()] + (0 if ch.isupper() else 8)
    return b


def case_start(case):
    """(squares uint8[90], turn, halfmove) of a golden case."""
    if case["start"] == "start":
        return start_position(), 1, 0
    return STARTS[case["start"]].copy(), case["turn"], case["halfmove"]


def case_order(case, move_from=None, move_to=None):
    """(move_rank uint16[2086] | None, type_rank list[8] | None) of a golden case: the `board.legal_moves` order it was
    generated with. ``order_seed``: a random permutation of the ids. ``order == "scan_desc_pawns_last"``: the scheme of
    bitboard libraries in the python-chess family -- non-pawn moves by from-square then to-square in DESCENDING square order,
    pawn moves after them -- which needs the major key by piece type (move_from / move_to: the action table's squares)."""
    if "order_seed" in case:
        return np.random.RandomState(case["order_seed"]).permutation(2086).astype(np.uint16), None
    if case.get("order") == "scan_desc_pawns_last":
        order = np.lexsort((-np.asarray(move_to, np.int64), -np.asarray(move_from, np.int64)))
        rank = np.empty(2086, np.uint16)
        rank[order] = np.arange(2086, dtype=np.uint16)
        return rank, [0, 1, 0, 0, 0, 0, 0, 0]
    return None, None
