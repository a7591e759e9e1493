#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by EXECUTING the reference's own tools.py and mcts.py.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

How the reference is executed: ``tools.py`` and ``mcts.py`` import the third-party
``cchess`` rules module, which is not installed and cannot be installed here. They only use three
names from it on this path (``cchess.Move.from_uci``, ``cchess.RED``, ``cchess.BLACK``), so an
empty placeholder module carrying those three names is registered and the reference's search code
runs unmodified on a duck-typed board. The board is backed by the CPU oracle's rules
(``oracle.OracleBoard``) and the evaluator is the deterministic ``oracle.evaluators`` hash, so the
vectors pin everything the reference's code computes -- action table, flip map, softmax, PUCT
arithmetic incl. NumPy dtype promotion, first-max tie-breaking, expansion order, backups, tree
reuse, pi, and the Dirichlet-mixed sampling under ``np.random.seed`` -- but NOT the rules
themselves (rules parity with cchess stays unpinned).

Outputs are data only (inputs + expected outputs); no reference source text is stored.
"""
from __future__ import annotations

import hashlib
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

sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as _oracle  # noqa: E402
from golden_cases import case_order  # noqa: E402
from oracle import OracleBoard  # noqa: E402
from oracle.evaluators import EVALUATORS  # noqa: E402

MOVE_FROM = [_oracle.lib().xq_move_from(i) for i in range(2086)]
MOVE_TO = [_oracle.lib().xq_move_to(i) for i in range(2086)]

REF = "/root/reference"


def load_reference():
    ph = types.ModuleType("cchess")
    ph.RED = True
    ph.BLACK = False

    class Move:  # only the constructor the search path uses (mcts.py:111)
        @staticmethod
        def from_uci(s):
            return s

    ph.Move = Move
    sys.modules["cchess"] = ph
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir("/tmp")  # tools.log would create ./logs next to the script otherwise
    import tools as ref_tools  # noqa
    import mcts as ref_mcts  # noqa
    os.chdir(cwd)
    silence_reference_log(ref_tools, ref_mcts)   # tools.log writes next to tools.py whatever the working directory is (tools.py:46-51)
    return ref_tools, ref_mcts


# endgame fixtures (piece codes: red = type, black = type + 8; types P1 C2 R3 N4 B5 A6 K7)
def _empty():
    return np.zeros(90, dtype=np.uint8)


def sq(name: str) -> int:
    return (ord(name[0]) - 97) + 9 * int(name[1])


def endgame_two_rooks():
    b = _empty()
    b[sq("d0")] = 7
    b[sq("a7")] = 3
    b[sq("b8")] = 3
    b[sq("e9")] = 15
    return b


def endgame_capture_to_bare():
    b = _empty()
    b[sq("e0")] = 7
    b[sq("d0")] = 6
    b[sq("e1")] = 9  # black pawn next to the red king: KxP leaves no attacking material
    b[sq("d9")] = 15
    b[sq("c9")] = 13
    return b


def wide_open_80_moves():
    b = _empty()
    for name, pc in {"e1": 7, "a2": 3, "i7": 3, "b4": 2, "h5": 2, "c3": 4, "g6": 4, "a6": 1, "c7": 1, "e6": 1, "g7": 1, "i6": 1,
                     "d0": 6, "f0": 6, "c0": 5, "g0": 5, "d9": 15, "e8": 14, "a9": 11}.items():
        b[sq(name)] = pc
    return b


def pawns_and_kings():
    b = _empty()
    for name, pc in {"d0": 7, "a3": 1, "c3": 1, "e3": 1, "g3": 1, "i3": 1, "f9": 15, "a6": 9, "c6": 9, "e6": 9, "g6": 9, "i6": 9}.items():
        b[sq(name)] = pc
    return b


def endgame_rook_knight():
    b = _empty()
    b[sq("e0")] = 7
    b[sq("e1")] = 6
    b[sq("c2")] = 4
    b[sq("h4")] = 3
    b[sq("d9")] = 15
    b[sq("e8")] = 14
    b[sq("a5")] = 11
    b[sq("g6")] = 9
    return b


CASES = [
    # name, start, turn, halfmove, evaluator, n_playout, plies, temps, seed, selfplay
    dict(name="start_hash_n50", start="start", ev="hash", n=50, plies=4, temps=[1.0, 1.0, 0.5, 0.5], seed=1, selfplay=True),
    dict(name="start_hash_n200", start="start", ev="hash", n=200, plies=3, temps=[1.0, 1.0, 0.5], seed=2, selfplay=True),
    dict(name="start_sharp_n400", start="start", ev="hash_sharp", n=400, plies=3, temps=[1.0, 0.5, 1.0], seed=3, selfplay=True),
    dict(name="start_sharp_n800", start="start", ev="hash_sharp", n=800, plies=2, temps=[1.0, 1.0], seed=4, selfplay=True),
    dict(name="start_uniform_n200", start="start", ev="uniform", n=200, plies=3, temps=[1.0, 1.0, 1.0], seed=5, selfplay=True),
    dict(name="start_hash_match_n100", start="start", ev="hash_sharp", n=100, plies=3, temps=[1e-3, 1e-3, 1e-3], seed=6, selfplay=False),
    dict(name="rooks_mate_n200", start="two_rooks", turn=1, halfmove=0, ev="hash", n=200, plies=3, temps=[1.0, 1.0, 1.0], seed=7, selfplay=True),
    dict(name="rooks_sixty_n300", start="two_rooks", turn=1, halfmove=114, ev="hash_sharp", n=300, plies=3, temps=[1.0, 1.0, 1.0], seed=8, selfplay=True),
    dict(name="bare_n150", start="capture_to_bare", turn=1, halfmove=3, ev="hash", n=150, plies=2, temps=[1.0, 1.0], seed=9, selfplay=True),
    dict(name="rookknight_black_n250", start="rook_knight", turn=0, halfmove=100, ev="hash_sharp", n=250, plies=4, temps=[1.0, 0.5, 1.0, 0.5], seed=10, selfplay=True),
    dict(name="rookknight_uniform_n120", start="rook_knight", turn=1, halfmove=0, ev="uniform", n=120, plies=3, temps=[1.0, 1.0, 1.0], seed=11, selfplay=True),
    dict(name="wide80_sharp_n300", start="wide80", turn=1, halfmove=0, ev="hash_sharp", n=300, plies=2, temps=[1.0, 0.5], seed=12, selfplay=True),
    dict(name="pawns_black_n400", start="pawns", turn=0, halfmove=0, ev="hash_sharp", n=400, plies=3, temps=[1.0, 1.0, 0.5], seed=13, selfplay=True),
    # `board.legal_moves` in a NON-canonical order (rank = RandomState(order_seed).permutation(2086)): children are inserted,
    # first-visited and tie-broken in that order (mcts.py:37-39,47-48,59-61); pins the engine's run-time move_rank table
    dict(name="start_sharp_shuffled_n200", start="start", ev="hash_sharp", n=200, plies=3, temps=[1.0, 1.0, 0.5], seed=14, selfplay=True, order_seed=77),
    dict(name="wide80_uniform_shuffled_n150", start="wide80", turn=1, halfmove=0, ev="uniform", n=150, plies=2, temps=[1.0, 1.0], seed=15, selfplay=True, order_seed=78),
    # the reference's CUDA path: under autocast the value is a float16 ndarray (net.py:178-189) and Node.value is then
    # accumulated in float16 (NEP 50); priors stay float32. Pins the engine's CCZ_FLAG_VALUE_F16 / the oracle's value_f16.
    dict(name="start_sharp_f16value_n300", start="start", ev="hash_sharp", n=300, plies=3, temps=[1.0, 1.0, 0.5], seed=16, selfplay=True, value_dtype="float16"),
    dict(name="rooks_f16value_n200", start="two_rooks", turn=1, halfmove=0, ev="hash", n=200, plies=3, temps=[1.0, 1.0, 1.0], seed=17, selfplay=True, value_dtype="float16"),
    # the reference's default search size (parameters.py:14 PLAYOUT = 1600) with tree reuse into a second move
    dict(name="start_sharp_n1600", start="start", ev="hash_sharp", n=1600, plies=2, temps=[1.0, 1.0], seed=18, selfplay=True),
    # a position-dependent order (major key = piece type of the mover): non-pawn moves by from / to square descending, pawn
    # moves last -- the iteration scheme of python-chess-style bitboard libraries; pins the engine's type_rank table
    dict(name="start_sharp_scanorder_n200", start="start", ev="hash_sharp", n=200, plies=3, temps=[1.0, 1.0, 0.5], seed=19, selfplay=True, order="scan_desc_pawns_last"),
    dict(name="wide80_scanorder_n150", start="wide80", turn=1, halfmove=0, ev="hash", n=150, plies=2, temps=[1.0, 1.0], seed=20, selfplay=True, order="scan_desc_pawns_last"),
]

STARTS = {"two_rooks": endgame_two_rooks, "capture_to_bare": endgame_capture_to_bare, "rook_knight": endgame_rook_knight,
          "wide80": wide_open_80_moves, "pawns": pawns_and_kings}


def make_board(case):
    if case["start"] == "start":
        return OracleBoard()
    return OracleBoard.from_array(STARTS[case["start"]](), case["turn"], case["halfmove"])


def main():
    ref_tools, ref_mcts = load_reference()
    out = {}
    meta = {"numpy": np.__version__, "cases": []}

    # ---- G1 action table, G2 flip map (tools.py:172-272, 133-166; collect.py:118-123)
    table = [ref_tools.move_id2move_action[i] for i in range(len(ref_tools.move_id2move_action))]
    assert len(table) == 2086
    for i, s in enumerate(table):
        assert ref_tools.move_action2move_id[s] == i
    with open(os.path.join(OUT, "action_table.txt"), "w") as f:
        f.write("\n".join(table) + "\n")
    meta["table_sha256"] = hashlib.sha256(",".join(table).encode()).hexdigest()
    flip_map = np.array([ref_tools.move_action2move_id[ref_tools.flip(table[i])] for i in range(2086)], dtype=np.int32)
    out["flip_map"] = flip_map
    meta["flip_sha256"] = hashlib.sha256(flip_map.astype("<i4").tobytes()).hexdigest()

    # ---- G4 softmax / pi from visit vectors (tools.py:126-129, mcts.py:165)
    rng = np.random.RandomState(12345)
    pis = []
    for k in (1, 2, 7, 44, 90):
        for temp in (1.0, 0.5, 1e-3):
            visits = rng.randint(0, 400, size=k)
            visits[rng.randint(0, k)] = 0
            if k > 1:
                visits[0] = max(1, visits[0])
            pi = ref_tools.softmax(1.0 / temp * np.log(np.array(visits) + 1e-10))
            pis.append((visits.astype(np.int32), temp, pi))
    for i, (v, t, p) in enumerate(pis):
        out[f"pi{i}_visits"] = v
        out[f"pi{i}_temp"] = np.float64(t)
        out[f"pi{i}_pi"] = p
    meta["n_pi"] = len(pis)

    # ---- G6 dtype facts established on the reference's Node
    root = ref_mcts.Node(None, 1.0)
    root.expand([(3, np.float32(0.25)), (9, np.float32(0.5))])
    c = root.children[3]
    c.update_recursive(-np.array([[0.3]], dtype=np.float32))
    facts = {
        "q_dtype_after_net_backup": str(np.asarray(c.value).dtype),
        "puct_dtype": str(np.asarray(c.puct_value(5)).dtype),
        "cpuct_times_prob_dtype": str(np.asarray(5 * c.prob).dtype),
        "sqrt_int_dtype": str(np.asarray(np.sqrt(root.visits)).dtype),
    }
    t = ref_mcts.Node(root, np.float32(0.1))
    t.parent = None
    t.update(-1.0)
    t.update(-1.0)
    facts["q_type_terminal_only"] = type(t.value).__name__
    t.update(np.array([[0.5]], dtype=np.float32))
    facts["q_dtype_mixed"] = str(np.asarray(t.value).dtype)
    unv = ref_mcts.Node(root, np.float32(0.1))
    facts["unvisited_is_inf"] = bool(unv.puct_value(5) == float("inf"))
    h = ref_mcts.Node(root, np.float32(0.1))
    h.update(-np.array([[0.3]], dtype=np.float16))
    facts["q_dtype_after_f16_backup"] = str(np.asarray(h.value).dtype)
    facts["puct_dtype_with_f16_q"] = str(np.asarray(h.puct_value(5)).dtype)
    h.update(1.0)
    facts["q_dtype_f16_then_terminal"] = str(np.asarray(h.value).dtype)
    meta["dtype_facts"] = facts

    # ---- G3/G5 search traces: the reference's MCTS_AI on the oracle-rules board
    for case in CASES:
        ev = EVALUATORS[case["ev"]]
        n_evals = [0]

        rank, trank = case_order(case, MOVE_FROM, MOVE_TO)

        vdt = np.dtype(case.get("value_dtype", "float32"))

        def policy(board, red_states=None, black_states=None, _ev=ev, _rank=rank, _trank=trank, _vdt=vdt):
            ids = board.legal_ids()
            if _rank is not None:  # what iterating a differently ordered `board.legal_moves` would hand to net.py:154-157
                sq_ = board.squares()
                ids = sorted(ids, key=lambda i: ((_trank[int(sq_[MOVE_FROM[i]]) & 7] if _trank else 0), int(_rank[i])))
            p, v = _ev(board.squares()[None, :], np.array([1 if board.turn else 0]))
            n_evals[0] += 1
            # shape/dtypes of net.py:202-205: float32 value on the CPU path, float16 on the CUDA (autocast) path
            return zip(ids, p[0][ids]), np.array([[v[0]]], dtype=np.float32).astype(_vdt)

        board = make_board(case)
        ai = ref_mcts.MCTS_AI(policy, c_puct=5, n_playout=case["n"], is_selfplay=case["selfplay"])
        np.random.seed(case["seed"])
        name = case["name"]
        plies_done = 0
        for ply in range(case["plies"]):
            if board.is_game_over() or board.is_tie():
                break
            temp = case["temps"][ply]
            # (re)run get_action's first half by hand to capture the tree before it is re-rooted,
            # then let get_action itself pick the move: same RNG consumption as the reference
            state = np.random.get_state()
            acts, probs = ai.mcts.get_move_probs(board, temp)
            kids = ai.mcts.root.children
            out[f"{name}_p{ply}_acts"] = np.array(acts, dtype=np.int32)
            out[f"{name}_p{ply}_visits"] = np.array([kids[a].visits for a in acts], dtype=np.int32)
            out[f"{name}_p{ply}_q"] = np.array([np.float32(np.asarray(kids[a].value).reshape(-1)[0]) for a in acts], dtype=np.float32)
            out[f"{name}_p{ply}_prior"] = np.array([np.float32(kids[a].prob) for a in acts], dtype=np.float32)
            out[f"{name}_p{ply}_pi"] = np.asarray(probs, dtype=np.float64)
            out[f"{name}_p{ply}_rootvisits"] = np.int64(ai.mcts.root.visits)
            # sampling exactly as mcts.py:216-229 does it
            np.random.set_state(state)
            if case["selfplay"]:
                dirichlet = np.random.dirichlet(0.2 * np.ones(len(probs)))
                mixed = 0.75 * probs + 0.25 * dirichlet
                move = np.random.choice(acts, p=mixed)
                out[f"{name}_p{ply}_mixed"] = mixed
                ai.mcts.update_with_move(move)
            else:
                move = np.random.choice(acts, p=probs)
                ai.mcts.update_with_move(-1)
            out[f"{name}_p{ply}_move"] = np.int32(move)
            board.push(ref_tools.move_id2move_action[int(move)])
            plies_done += 1
        out[f"{name}_final_sq"] = board.squares()
        meta["cases"].append({**{k: v for k, v in case.items()}, "plies_done": plies_done, "evals": n_evals[0]})
        print(name, "plies", plies_done, "evals", n_evals[0])

    # ---- G7: decode_board (tools.py:74-106) under ANOTHER piece-type numbering of the rules module: the duck-typed board
    # reports piece_type = 8 - t (KING 1 .. PAWN 7); pins the engine's run-time plane_of_type table (channel = type_ref - 1)
    class _AltPiece:
        def __init__(self, pc):
            self.piece_type = 8 - (pc & 7)
            self.color = not bool(pc & 8)

    class _AltBoard:
        def __init__(self, sq):
            self._sq = sq

        def piece_at(self, s):
            pc = int(self._sq[s])
            return _AltPiece(pc) if pc else None

    for nm, sqs in (("start", OracleBoard().squares()), ("wide80", wide_open_80_moves())):
        red, black = ref_tools.decode_board(_AltBoard(sqs))
        out[f"decode_alt_{nm}_sq"] = np.asarray(sqs, dtype=np.uint8)
        out[f"decode_alt_{nm}_red"] = np.asarray(red)
        out[f"decode_alt_{nm}_black"] = np.asarray(black)
    meta["decode_alt_plane_of_type"] = [0, 6, 5, 4, 3, 2, 1, 0]

    # ---- G5b: one end-to-end MCTS_AI.get_action call (return_prob=True), nothing done by hand
    ev = EVALUATORS["hash_sharp"]

    def policy2(board, red_states=None, black_states=None):
        ids = board.legal_ids()
        p, v = ev(board.squares()[None, :], np.array([1 if board.turn else 0]))
        return zip(ids, p[0][ids]), np.array([[v[0]]], dtype=np.float32)

    board = OracleBoard()
    ai = ref_mcts.MCTS_AI(policy2, c_puct=5, n_playout=120, is_selfplay=True)
    np.random.seed(2024)
    calls = []
    for ply in range(3):
        move, move_probs = ai.get_action(board, temp=1.0, return_prob=True, on_playout=lambda d: calls.append(d))
        out[f"getaction_p{ply}_move"] = np.int32(move)
        out[f"getaction_p{ply}_probs"] = move_probs
        board.push(ref_tools.move_id2move_action[int(move)])
    meta["on_playout_calls"] = calls[:200]

    # ---- the reference's module-level constants (parameters.py:1-28)
    import parameters as ref_params  # noqa
    meta["parameters"] = {k: getattr(ref_params, k) for k in dir(ref_params) if k.isupper()}

    np.savez_compressed(os.path.join(OUT, "reference_search.npz"), **out)
    with open(os.path.join(OUT, "reference_search.json"), "w") as f:
        json.dump(meta, f, indent=1, default=lambda o: o if not isinstance(o, np.generic) else o.item())
    print("table sha", meta["table_sha256"])
    print("facts", facts)


if __name__ == "__main__":
    main()
    assert_reference_untouched()
