#!/usr/bin/env python3
"""probe_cchess.py -- turn "rules parity with cchess is unpinned" into ONE command for someone who HAS the module.

The reference takes its rules from the third-party module ``cchess`` (GitHub windshadow233/python-chinese-chess, un-pinned and
un-vendored: reference README.md:21, .gitignore:3). It is not in the build image and cannot be installed there, so this build's
rules are its own statement of Xiangqi, with every choice that ``cchess`` could make differently exposed as
a run-time table or flag. This script asks a real ``cchess`` for those choices:

    pip install <python-chinese-chess as the reference's README says>     (never in the build container, never on a GPU box)
    python tools/probe_cchess.py --out rules_probe/                        (seconds; pure Python, needs numpy, no GPU, no torch)

and writes

  * ``preset.json`` -- ``plane_of_type`` (the PIECE_TYPES numbering, call site tools.py:100), ``move_rank`` / ``type_rank`` (the
    iteration order of ``board.legal_moves``: net.py:154-157 -> mcts.py:37-39,47-48,59-61), ``pawn_move_resets_clock``,
    ``perpetual_check`` (game end and winner: mcts.py:116-126, game.py:208-219, tools.py:119-123), the raw facts they were derived
    from, and ``unsupported_differences``: behaviours of this cchess that no table of this build expresses (each one is a real
    parity gap to fix in the kernels and the oracle, not to paper over);
  * ``cchess_golden.npz`` -- ~200 positions (seeded random walks + crafted endings) with what cchess said about each: legal moves
    IN ITS ORDER, is_game_over / the three draw predicates / the winner. ``tests/test_oracle_rules.py`` replays such a file against
    the oracle with the preset installed, ``tests/test_gpu_rule_tables.py`` against the kernels.

Consumers: ``chinesechesszero_amd.tools.set_rules(preset="rules_probe/preset.json")`` (every engine created afterwards takes the
tables: ``ccz_config.move_rank_host / type_rank / plane_of_type / rule_flags``) and ``oracle.set_rules_from_file(...)``.

The module under test is a parameter (``probe(cchess_module)``): the repository's tests run the whole probe against this build's
own rules duck-typed as ``cchess`` and require the round trip to reproduce the preset that was installed.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NMOVES = 2086
# this build's piece-type codes (include/cczero.h): the index of plane_of_type / type_rank
PAWN, CANNON, ROOK, KNIGHT, BISHOP, ADVISOR, KING = 1, 2, 3, 4, 5, 6, 7
TYPE_NAMES = {PAWN: "pawn", CANNON: "cannon", ROOK: "rook", KNIGHT: "knight", BISHOP: "bishop", ADVISOR: "advisor", KING: "king"}
FEN_LETTER = {PAWN: "p", CANNON: "c", ROOK: "r", KNIGHT: "n", BISHOP: "b", ADVISOR: "a", KING: "k"}
START_FEN = "rnbakabnr/9/1c5c1/p1p1p1p1p/9/9/P1P1P1P1P/1C5C1/9/RNBAKABNR w - - 0 1"
# where the start position holds one piece of each type (red side): square = file + 9 * rank (reference tools.py:91)
START_SQUARES = {ROOK: 0, KNIGHT: 1, BISHOP: 2, ADVISOR: 3, KING: 4, CANNON: 9 * 2 + 1, PAWN: 9 * 3 + 0}


def action_table():
    """The reference's 2086 move strings in id order (tests/golden/action_table.txt: printed by its own tools.py:172-272)."""
    with open(os.path.join(ROOT, "tests", "golden", "action_table.txt")) as f:
        names = f.read().split()
    assert len(names) == NMOVES
    return names, {s: i for i, s in enumerate(names)}


def sq_of(name: str) -> int:
    return (ord(name[0]) - 97) + 9 * int(name[1])


def fen_of(pieces: dict, turn_red: bool, halfmove: int = 0, fullmove: int = 1) -> str:
    """pieces: {square: (build type, is_red)} -> Xiangqi FEN (rank 9 first)."""
    rows = []
    for rank in range(9, -1, -1):
        row, gap = "", 0
        for file in range(9):
            pc = pieces.get(file + 9 * rank)
            if pc is None:
                gap += 1
                continue
            if gap:
                row += str(gap)
                gap = 0
            ch = FEN_LETTER[pc[0]]
            row += ch.upper() if pc[1] else ch
        rows.append(row + (str(gap) if gap else ""))
    return "/".join(rows) + f" {'w' if turn_red else 'b'} - - {halfmove} {fullmove}"


def P(**kw):
    """{'e0': 'K', 'a9': 'r', ...}: upper case = red."""
    inv = {v: k for k, v in FEN_LETTER.items()}
    return {sq_of(k): (inv[v.lower()], v.isupper()) for k, v in kw.items()}


class Adapter:
    """The handful of cchess calls the reference makes, behind try/except: what is missing is recorded, not fatal."""

    def __init__(self, mod):
        self.m = mod
        self.notes = []

    def board(self, fen=None):
        if fen is None:
            return self.m.Board()
        try:
            return self.m.Board(fen)
        except Exception:
            b = self.m.Board()
            b.set_fen(fen)
            return b

    def moves(self, b):
        out = []
        for mv in b.legal_moves:
            out.append(mv.uci() if hasattr(mv, "uci") else str(mv))
        return out

    def push(self, b, uci):
        b.push(self.m.Move.from_uci(uci))

    def red_is(self):
        return getattr(self.m, "RED", True)

    def halfmove(self, b):
        for name in ("halfmove_clock", "halfmove"):
            if hasattr(b, name):
                return int(getattr(b, name))
        try:
            return int(b.fen().split()[4])
        except Exception:
            return None

    def winner(self, b):
        """'red' / 'black' / None (draw or no outcome) and the termination text."""
        try:
            o = b.outcome()
        except Exception as e:   # (mcts.py:125 would raise here too)
            return "error", repr(e)
        if o is None:
            return "none", ""
        w = getattr(o, "winner", None)
        term = str(getattr(o, "termination", ""))
        if w is None:
            return None, term
        return ("red" if w == self.red_is() else "black"), term

    def flags(self, b):
        d = {}
        for name in ("is_game_over", "is_insufficient_material", "is_fourfold_repetition", "is_sixty_moves", "is_check", "is_checkmate", "is_stalemate"):
            try:
                d[name] = bool(getattr(b, name)())
            except Exception as e:
                d[name] = None
                self.notes.append(f"{name}: {e!r}")
        return d


def read_position(ad, b, numbering):
    """(squares uint8 [90] in this build's piece codes, turn 1 = RED)."""
    inv = {v: k for k, v in numbering.items()}
    sq = np.zeros(90, np.uint8)
    for i in range(90):
        pc = b.piece_at(i)
        if pc:
            t = inv[int(pc.piece_type)]
            sq[i] = t if pc.color == ad.red_is() else t + 8
    return sq, 1 if b.turn == ad.red_is() else 0


def probe_numbering(ad):
    """PIECE_TYPES as cchess numbers them, read off the start position (no constant names assumed)."""
    b = ad.board()
    num = {}
    for t, sq in START_SQUARES.items():
        pc = b.piece_at(sq)
        if pc is None or pc.color != ad.red_is():
            raise RuntimeError(f"start position: expected a red {TYPE_NAMES[t]} on square {sq}; this cchess orients the board differently")
        num[t] = int(pc.piece_type)
    if sorted(num.values()) != list(range(1, 8)):
        raise RuntimeError(f"piece types are not 1..7: {num}")
    return num


def walk_positions(ad, names, n_games=8, plies=26, seed=20240611):
    """Seeded random walks from the opening. The move is chosen among the SORTED move strings, so the positions do not depend on
    the order under test. Returns [(fen-independent record)]: moves so far are replayable with any rules implementation."""
    rs = np.random.RandomState(seed)
    games = []
    for _ in range(n_games):
        b = ad.board()
        line = []
        for _ply in range(plies):
            order = ad.moves(b)
            if not order or ad.flags(b)["is_game_over"]:
                break
            games.append((list(line), order))
            mv = sorted(order)[int(rs.randint(len(order)))]
            ad.push(b, mv)
            line.append(mv)
    return games


# ---------------------------------------------------------------- the order of board.legal_moves
def parametric_orders(names):
    fr = np.array([sq_of(s[:2]) for s in names])
    to = np.array([sq_of(s[2:]) for s in names])
    ident = np.arange(NMOVES)
    cands = {"id ascending": ident, "id descending": -ident,
             "from ascending, to ascending": fr * 100 + to, "from descending, to descending": -(fr * 100 + to),
             "from ascending, to descending": fr * 100 - to, "from descending, to ascending": -fr * 100 + to,
             "to ascending, from ascending": to * 100 + fr, "to descending, from descending": -(to * 100 + fr)}
    out = {}
    for k, key in cands.items():
        order = np.argsort(key, kind="stable")
        rank = np.empty(NMOVES, np.int64)
        rank[order] = ident
        out[k] = rank
    return out


def toposort(n, edges):
    """Kahn's algorithm with the smallest free index first; None if the precedence graph has a cycle."""
    import heapq
    succ = [[] for _ in range(n)]
    indeg = [0] * n
    for a, b in set(edges):
        succ[a].append(b)
        indeg[b] += 1
    heap = [i for i in range(n) if indeg[i] == 0]
    heapq.heapify(heap)
    out = []
    while heap:
        i = heapq.heappop(heap)
        out.append(i)
        for j in succ[i]:
            indeg[j] -= 1
            if indeg[j] == 0:
                heapq.heappush(heap, j)
    return out if len(out) == n else None


def fit_order(samples, names):
    """samples: [(ids in cchess order, mover type of each)]. Finds (type_rank, move_rank) with order = ascending
    (type_rank[type], move_rank[id]) -- what ccz_config expresses -- or says why not."""
    rep = {"positions": len(samples), "moves_seen": len({i for ids, _ in samples for i in ids})}

    def consistent(rank, trank):
        for ids, types in samples:
            keys = [(trank[t], rank[i]) for i, t in zip(ids, types)]
            if any(keys[j] >= keys[j + 1] for j in range(len(keys) - 1)):
                return False
        return True

    # 1. the mover's type as a major key. Types that interleave (a before b here, b before a there) share a class; between classes
    #    the order must be one-way. (python-chess style generators: every non-pawn piece in one square scan, pawn moves after.)
    id_edges = [(a, b) for ids, _ in samples for a, b in zip(ids, ids[1:])]
    trank = [0] * 8
    type_major = False
    if toposort(NMOVES, id_edges) is None:   # the same id is early in one position and late in another: not a static permutation
        reach = [[i == j for j in range(8)] for i in range(8)]
        for _, types in samples:
            for ta, tb in zip(types, types[1:]):
                reach[ta][tb] = True
        for m in range(8):
            for i in range(8):
                for j in range(8):
                    reach[i][j] = reach[i][j] or (reach[i][m] and reach[m][j])
        present = sorted({t for _, ts in samples for t in ts})
        classes = []
        for t in present:
            for c in classes:
                if reach[t][c[0]] and reach[c[0]][t]:
                    c.append(t)
                    break
            else:
                classes.append([t])
        import functools
        classes.sort(key=functools.cmp_to_key(lambda x, y: -1 if (reach[x[0]][y[0]] and not reach[y[0]][x[0]]) else
                                              (1 if (reach[y[0]][x[0]] and not reach[x[0]][y[0]]) else x[0] - y[0])))
        if len(classes) < 2:
            rep["fit"] = "none: neither a static permutation of the ids nor a type-major order describes this sequence"
            return None, None, rep
        for rank_, c in enumerate(classes):
            for t in c:
                trank[t] = rank_
        type_major = True
        id_edges = [(a, b) for ids, types in samples for (a, ta), (b, tb) in zip(zip(ids, types), list(zip(ids, types))[1:]) if trank[ta] == trank[tb]]
    # 2. a closed-form minor order that agrees with every observation extrapolates to the ids never seen
    for name, rank in parametric_orders(names).items():
        if consistent(rank, trank):
            rep["fit"] = ("type-major (" + ", ".join(f"{TYPE_NAMES[t]}:{trank[t]}" for t in range(1, 8)) + "), then " if type_major else "") + name
            rep["extrapolated"] = True
            return (trank if type_major else None), (None if (name == "id ascending") else rank.astype(np.uint16)), rep
    order = toposort(NMOVES, id_edges)
    if order is None:
        rep["fit"] = "none: the minor order is not a static permutation either"
        return None, None, rep
    rank = np.empty(NMOVES, np.uint16)
    rank[np.asarray(order)] = np.arange(NMOVES, dtype=np.uint16)
    rep["fit"] = ("type-major, then " if type_major else "") + "a permutation fitted to the observed pairs (no closed form found)"
    rep["extrapolated"] = False   # ids never observed sit where the smallest-index-first topological sort put them
    return (trank if type_major else None), rank, rep


# ---------------------------------------------------------------- crafted endings
def crafted(ad):
    """Positions that separate the readings of the game-end rules. Returns (facts, [(label, fen, moves)]) -- every position is
    also a golden record."""
    facts, recs = {}, []
    base = dict(d0="K", f9="k")

    def play(label, pieces, turn_red, moves=(), halfmove=0):
        fen = fen_of(P(**pieces), turn_red, halfmove, 1 + halfmove // 2)
        b = ad.board(fen)
        for mv in moves:
            ad.push(b, mv)
        recs.append((label, fen, list(moves)))
        return b

    # (a) no legal move, not in check (Xiangqi: the side to move LOSES; python-chess lineage: stalemate is a draw)
    b = play("stalemate", dict(e0="K", e9="k", d9="r", f9="r", a1="r", e5="p"), True)
    facts["stalemate"] = {"legal_moves": len(ad.moves(b)), **ad.flags(b), "winner": ad.winner(b)[0], "termination": ad.winner(b)[1]}
    # (b) checkmate
    b = play("checkmate", dict(e0="K", e9="k", d9="r", f9="r", a1="r", e3="r"), True)
    facts["checkmate"] = {"legal_moves": len(ad.moves(b)), **ad.flags(b), "winner": ad.winner(b)[0]}
    # (c) the sixty-move clock: 119 plies without a capture, one quiet move more
    rooks = dict(base, a0="R", i9="r")
    b = play("clock_119", rooks, True, halfmove=119)
    before = ad.flags(b)["is_sixty_moves"]
    ad.push(b, "a0a1")
    recs.append(("clock_120", recs[-1][1], ["a0a1"]))
    facts["sixty_moves"] = {"at_119": before, "at_120": ad.flags(b)["is_sixty_moves"], "game_over_at_120": ad.flags(b)["is_game_over"],
                            "winner_at_120": ad.winner(b)[0], "clock_after": ad.halfmove(b)}
    b = play("pawn_move_clock", dict(base, a0="R", i9="r", a3="P"), True, halfmove=50)
    ad.push(b, "a3a4")
    facts["pawn_move"] = {"clock_before": 50, "clock_after": ad.halfmove(b)}
    b = play("capture_clock", dict(base, a0="R", a9="r"), True, halfmove=50)
    ad.push(b, "a0a9")
    facts["capture"] = {"clock_before": 50, "clock_after": ad.halfmove(b)}
    # (d) repetition without checks: how many occurrences make is_fourfold_repetition() true
    b = play("repetition_quiet", rooks, True)
    cyc, first, line = ["a0a1", "i9i8", "a1a0", "i8i9"], None, []
    for n in range(2, 8):
        for mv in cyc:
            ad.push(b, mv)
            line.append(mv)
        f = ad.flags(b)
        if first is None and f["is_fourfold_repetition"]:
            first = n
            facts["repetition"] = {"occurrences_needed": n, "game_over": f["is_game_over"], "winner": ad.winner(b)[0], "termination": ad.winner(b)[1]}
            recs.append(("repetition_quiet_end", recs[-1][1] if recs[-1][0] == "repetition_quiet" else fen_of(P(**rooks), True), list(line)))
            break
    if first is None:
        facts["repetition"] = {"occurrences_needed": None}
    # (e) perpetual check: red checks with every move of the cycle, black only steps aside
    pc = dict(d0="K", e9="k", a8="R")
    b = play("perpetual_check", pc, True)
    cyc, line, done = ["a8a9", "e9e8", "a9a8", "e8e9"], [], None
    for n in range(2, 8):
        for mv in cyc:
            ad.push(b, mv)
            line.append(mv)
        f = ad.flags(b)
        if f["is_game_over"] or f["is_fourfold_repetition"]:
            w, term = ad.winner(b)
            done = {"occurrences": n, "game_over": f["is_game_over"], "is_fourfold_repetition": f["is_fourfold_repetition"], "winner": w, "termination": term}
            recs.append(("perpetual_check_end", fen_of(P(**pc), True), list(line)))
            break
    facts["perpetual_check"] = done or {"occurrences": None}
    # (f) insufficient material: bare kings, and one extra red piece of every type
    ins = {}
    for label, extra in (("kings", {}), ("advisor", dict(e1="A")), ("bishop", dict(c0="B")), ("pawn", dict(a6="P")), ("knight", dict(b0="N")),
                         ("cannon", dict(b2="C")), ("rook", dict(a0="R")), ("advisors_bishops_both", dict(e1="A", c0="B", e8="a", c9="b"))):
        b = play("material_" + label, dict(base, **extra), True)
        f = ad.flags(b)
        ins[label] = {"is_insufficient_material": f["is_insufficient_material"], "is_game_over": f["is_game_over"]}
    facts["insufficient_material"] = ins
    return facts, recs


def derive_flags(facts):
    """The rule flags this build has, and every probed behaviour it has NO switch for."""
    diff = []
    st = facts["stalemate"]
    if not (st["legal_moves"] == 0 and st["is_game_over"] and st["winner"] == "black"):
        diff.append(f"no legal move without check: this build scores a LOSS for the side to move (winner black in the probe); cchess says {st}")
    cm = facts["checkmate"]
    if not (cm["is_game_over"] and cm["winner"] == "black"):
        diff.append(f"checkmate probe: expected game over, winner black; cchess says {cm}")
    sm = facts["sixty_moves"]
    if not (sm["at_119"] is False and sm["at_120"] is True and sm["game_over_at_120"]):
        diff.append(f"sixty-move rule: this build draws at 120 plies without a capture when a legal move exists; cchess says {sm}")
    pawn_resets = facts["pawn_move"]["clock_after"] == 0
    if facts["capture"]["clock_after"] != 0:
        diff.append(f"a capture does not reset the clock: {facts['capture']}")
    rp = facts["repetition"]
    if rp.get("occurrences_needed") != 4:
        diff.append(f"is_fourfold_repetition(): this build needs 4 occurrences of the position (pieces + side to move); cchess needs {rp.get('occurrences_needed')}")
    elif not rp.get("game_over"):
        diff.append("a fourfold repetition does not end the game in cchess (is_game_over() false): mcts.py:116-126 would then call outcome() on a live game")
    pcf = facts["perpetual_check"]
    perpetual = bool(pcf.get("winner") == "black")      # the checking side (red) loses
    if pcf.get("occurrences") is None:
        diff.append("the perpetual-check line never ended the game")
    elif pcf.get("winner") == "red":
        diff.append(f"perpetual check is WON by the checking side in cchess: {pcf}")
    ins = facts["insufficient_material"]
    want = {"kings": True, "advisor": True, "bishop": True, "advisors_bishops_both": True, "pawn": False, "knight": False, "cannon": False, "rook": False}
    got = {k: v["is_insufficient_material"] for k, v in ins.items()}
    if got != want:
        diff.append(f"is_insufficient_material(): this build = neither side has a rook, knight, cannon or pawn; cchess says {got}")
    if not ins["kings"]["is_game_over"]:
        diff.append("insufficient material does not end the game in cchess (is_game_over() false)")
    return pawn_resets, perpetual, diff


def probe(cchess_module, out_dir=None, n_games=8, plies=26, seed=20240611):
    names, ids_of = action_table()
    ad = Adapter(cchess_module)
    numbering = probe_numbering(ad)                        # build type -> cchess piece_type
    plane_of_type = [0] * 8
    for t, ct in numbering.items():
        plane_of_type[t] = ct - 1                          # channel = piece_type - 1 (reference tools.py:100)
    # ---- legal_moves order on ~200 walk positions
    games = walk_positions(ad, names, n_games, plies, seed)
    samples, golden = [], []
    for line, order in games:
        b = ad.board()
        for mv in line:
            ad.push(b, mv)
        sq, turn = read_position(ad, b, numbering)
        unknown = [m for m in order if m not in ids_of]
        if unknown:
            raise RuntimeError(f"cchess produced moves outside the reference's 2086-move table: {unknown[:4]}")
        ids = [ids_of[m] for m in order]
        samples.append((ids, [int(sq[sq_of(names[i][:2])]) & 7 for i in ids]))
        w = ad.winner(b)[0]
        golden.append(("walk", START_FEN, line, sq, turn, ad.halfmove(b), ids, ad.flags(b), w))
    type_rank, move_rank, order_report = fit_order(samples, names)
    # ---- crafted endings
    facts, recs = crafted(ad)
    for label, fen, moves in recs:
        b = ad.board(fen)
        for mv in moves:
            ad.push(b, mv)
        sq, turn = read_position(ad, b, numbering)
        order = ad.moves(b)
        golden.append((label, fen, moves, sq, turn, ad.halfmove(b), [ids_of[m] for m in order], ad.flags(b), ad.winner(b)[0]))
    pawn_resets, perpetual, diff = derive_flags(facts)
    if order_report.get("fit", "").startswith("none"):
        diff.append("legal_moves order: " + order_report["fit"])
    src = getattr(cchess_module, "__file__", None)
    digest = None
    if src and os.path.exists(src):
        with open(src, "rb") as f:
            digest = hashlib.sha256(f.read()).hexdigest()
    preset = {"schema": 1, "what": "rule profile of a cchess module, probed by tools/probe_cchess.py; install with chinesechesszero_amd.tools.set_rules(preset=<this file>)",
              "cchess": {"file": src, "sha256_of_init": digest, "version": str(getattr(cchess_module, "__version__", "unknown"))},
              "piece_type_numbering": {TYPE_NAMES[t]: n for t, n in numbering.items()},
              "plane_of_type": plane_of_type, "type_rank": type_rank, "move_rank": None if move_rank is None else [int(x) for x in move_rank],
              "pawn_move_resets_clock": bool(pawn_resets), "perpetual_check": bool(perpetual),
              "legal_moves_order": order_report, "facts": facts, "adapter_notes": sorted(set(ad.notes)),
              "unsupported_differences": diff}
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "preset.json"), "w") as f:
            json.dump(preset, f, indent=1)
        save_golden(os.path.join(out_dir, "cchess_golden.npz"), golden)
    return preset, golden


def save_golden(path, golden):
    n = len(golden)
    ids = np.full((n, 128), -1, np.int16)
    k = np.zeros(n, np.int32)
    sq = np.zeros((n, 90), np.uint8)
    turn = np.zeros(n, np.uint8)
    half = np.full(n, -1, np.int32)
    flags = np.zeros((n, 4), np.int8)      # is_game_over, is_insufficient_material, is_fourfold_repetition, is_sixty_moves (-1 = unknown)
    winner = np.full(n, -2, np.int8)       # 1 red, 0 black, -1 draw, -2 no outcome / game not over
    meta = []
    for j, (label, fen, moves, s, t, h, order, fl, w) in enumerate(golden):
        k[j] = len(order)
        ids[j, :len(order)] = order
        sq[j], turn[j] = s, t
        half[j] = -1 if h is None else h
        for c, name in enumerate(("is_game_over", "is_insufficient_material", "is_fourfold_repetition", "is_sixty_moves")):
            flags[j, c] = -1 if fl[name] is None else int(fl[name])
        winner[j] = {"red": 1, "black": 0, None: -1}.get(w, -2)
        meta.append({"label": label, "fen": fen, "moves": moves})
    np.savez_compressed(path, ids=ids, k=k, squares=sq, turn=turn, halfmove=half, flags=flags, winner=winner, meta=json.dumps(meta))


def load_preset(path):
    """preset.json -> keyword arguments of tools.set_rules / oracle.set_rules (None entries = this build's defaults)."""
    with open(path) as f:
        p = json.load(f)
    if p.get("schema") != 1:
        raise ValueError(f"{path}: not a schema-1 rule preset")
    mr = p.get("move_rank")
    return dict(move_rank=None if mr is None else np.asarray(mr, np.uint16), plane_of_type=tuple(p["plane_of_type"]),
                type_rank=None if p.get("type_rank") is None else tuple(p["type_rank"]),
                pawn_move_resets_clock=bool(p.get("pawn_move_resets_clock")), perpetual_check=bool(p.get("perpetual_check"))), p


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--out", default="rules_probe", help="directory for preset.json and cchess_golden.npz")
    ap.add_argument("--games", type=int, default=8)
    ap.add_argument("--plies", type=int, default=26)
    a = ap.parse_args()
    try:
        import cchess
    except ImportError:
        print("probe_cchess: the module `cchess` (python-chinese-chess, reference README.md:21) is not installed here -- "
              "run this where the reference runs; nothing was written", file=sys.stderr)
        return 2
    preset, golden = probe(cchess, a.out, a.games, a.plies)
    print(f"wrote {a.out}/preset.json and {a.out}/cchess_golden.npz ({len(golden)} positions)")
    print("piece types:", preset["piece_type_numbering"], "-> plane_of_type", preset["plane_of_type"])
    print("legal_moves order:", preset["legal_moves_order"]["fit"])
    print("pawn_move_resets_clock:", preset["pawn_move_resets_clock"], " perpetual_check:", preset["perpetual_check"])
    if preset["unsupported_differences"]:
        print("\nBEHAVIOURS THIS BUILD HAS NO SWITCH FOR (real parity gaps):")
        for d in preset["unsupported_differences"]:
            print("  -", d)
        return 1
    print("every probed behaviour is expressible by the preset: install it with tools.set_rules(preset=...) and re-run the rule tests")
    return 0


if __name__ == "__main__":
    sys.exit(main())
