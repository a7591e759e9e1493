"""GPU: the run-time rule tables of ccz_config (ABI 2) -- `legal_moves` order and piece-type -> plane map --, the
fresh-root entry point ccz_reset_tree, and the split error bits. The two tables are the choices no golden trace can
pin against the absent cchess module: they must be switchable without touching a kernel."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine(B, n, **kw):
    from chinesechesszero_amd.engine import SelfPlayEngine
    # parity tests: a pruned subtree or an adjudicated game is an error, not a counter (CCZ_FLAG_STRICT) -- unless the test sets a ply cap
    # or a node budget on purpose
    kw.setdefault("strict", not ({"max_plies", "max_nodes", "reserve_nodes"} & set(kw)))
    return SelfPlayEngine(B, n_playout=n, **kw)


def test_shuffled_legal_move_order_matches_oracle_bit_for_bit(rules_of_case):
    """6 boards, 3 plies x 80 sims with a shuffled `legal_moves` order: the leaf's id list (expansion order), first-visit
    order, PUCT tie-breaks and hence N/Q/P equal the oracle's under the same rank table; and differ from ascending ids."""
    from gpu_harness import Lockstep
    from oracle import OracleBoard
    B, n = 6, 80
    rank = rules_of_case({"order_seed": 4242})
    e = _engine(B, n, seed=9, move_rank=rank)
    ls = Lockstep(e, [OracleBoard() for _ in range(B)], kind="hash_sharp", salts=[31, 32, 33, 34, 35, 36])
    first_acts = None
    for ply in range(3):
        if ply == 1:
            ls.run_fused(n, check_leaf=True)
        else:
            for _ in range(n):
                ls.step(check_leaf=True)   # leaf ids are compared with oracle.legal_ids() (rank order) at every step
        rc = ls.compare_roots()
        if first_acts is None:
            first_acts = rc["acts"][0][:rc["k"][0]].astype(int).tolist()
        pi = e.root_pi(temps=1.0)
        ls.play([int(rc["acts"][b][int(np.argmax(pi[b][:rc["k"][b]]))]) for b in range(B)])
    e.check_healthy()
    assert sorted(first_acts) != first_acts and [int(rank[a]) for a in first_acts] == sorted(int(rank[a]) for a in first_acts)
    # uniform priors: every PUCT comparison is an exact tie, so the visit pattern IS the order
    e2 = _engine(1, 50, move_rank=rank)
    ls2 = Lockstep(e2, [OracleBoard()], kind="uniform")
    for _ in range(50):
        ls2.step(check_leaf=True)
    rc2 = ls2.compare_roots()
    assert rc2["visits"][0][:44].tolist() == [2] * 5 + [1] * 39  # 49 descents over 44 children in insertion order
    with pytest.raises(Exception, match="permutation"):
        _engine(1, 4, move_rank=np.zeros(2086, np.uint16))


def test_plane_map_switches_the_encoding_everywhere(golden):
    """plane_of_type (channel of a piece type, tools.py:100) reaches the evaluator input of the search path AND the
    harvested training states; checked against the reference's own decode_board run under that numbering (golden G7)."""
    import oracle
    from gpu_harness import Lockstep
    from oracle import OracleBoard
    pot = golden["meta"]["decode_alt_plane_of_type"]
    d = golden["data"]
    try:
        oracle.set_rules(plane_of_type=pot)
        e = _engine(2, 8, seed=1, plane_of_type=pot, max_plies=3)
        e.set_position(1, d["decode_alt_wide80_sq"], 1, 0)
        leaf = e.select_leaves().float().cpu().numpy()
        assert np.array_equal(leaf[0][7], d["decode_alt_start_red"]) and np.array_equal(leaf[0][15], d["decode_alt_start_black"])
        assert np.array_equal(leaf[1][7], d["decode_alt_wide80_red"]) and np.array_equal(leaf[1][15], d["decode_alt_wide80_black"])
        boards = [OracleBoard(), OracleBoard.from_array(d["decode_alt_wide80_sq"], 1, 0)]
        ls = Lockstep(e, boards, kind="hash", salts=[1, 2])
        for ply in range(3):
            for _ in range(8):
                ls.step(check_leaf=True)      # planes compared with oracle.leaf_planes() under the same map
            rc = ls.compare_roots()
            ls.play([int(rc["acts"][b][0]) for b in range(2)])
        for _ in range(8):
            ls.step(check_leaf=True)
        e.finish_move()                        # ply cap 3 -> both games adjudicated
        assert e.game_status()["over"].all()
        states, pi, z = e.harvest()
        states = states.cpu().numpy()
        assert states.shape[0] == 2 * 3 * 2
        assert np.array_equal(states[0][0], d["decode_alt_start_red"]) and np.array_equal(states[0][8], d["decode_alt_start_black"])
        assert np.array_equal(states[6][0], d["decode_alt_wide80_red"]) and np.array_equal(states[6][8], d["decode_alt_wide80_black"])
        e.check_healthy()
    finally:
        oracle.set_rules()
    e0 = _engine(1, 4)
    assert not np.array_equal(e0.select_leaves().float().cpu().numpy()[0][7], d["decode_alt_start_red"])  # default map differs
    with pytest.raises(Exception, match="permutation"):
        _engine(1, 4, plane_of_type=[0, 1, 1, 2, 3, 4, 5, 6])


def test_host_mirror_follows_installed_rules(rules_of_case):
    """tools.set_rules: Board.legal_ids / legal_moves iterate in the installed order, MCTS engines are created with it, and
    a reference-style policy_value_fn therefore sees (and expands) the moves in that order."""
    from chinesechesszero_amd import tools
    from chinesechesszero_amd.game import Board
    from chinesechesszero_amd.mcts import MCTS
    from oracle import OracleBoard
    from oracle.evaluators import hash_eval
    rank = rules_of_case({"order_seed": 99}, product=True)
    b = Board()
    ob = OracleBoard()
    assert b.legal_ids() == ob.legal_ids() and [m.uci() for m in b.legal_moves] == ob.legal_moves
    seen = []

    def policy(board, red_states=None, black_states=None):
        ids = board.legal_ids()
        seen.append(list(ids))
        p, v = hash_eval(board.squares()[None, :], np.array([1 if board.turn else 0]), salt=5, scale=40.0)
        return zip(ids, p[0][ids]), np.array([[v[0]]], dtype=np.float32)

    m = MCTS(policy, c_puct=5, n_playout=30)
    acts, probs = m.get_move_probs(b, temp=1.0)
    assert list(acts) == ob.legal_ids() and seen[0] == ob.legal_ids()
    assert tools.MOVE_RANK is not None and np.array_equal(m._engine.move_rank, rank)


def test_reset_tree_keeps_position_history_and_record():
    """ccz_reset_tree == MCTS.update_with_move(-1) (mcts.py:176-178): fresh root, everything else stays -- in particular
    the repetition history, which a set_position-based reset would lose."""
    from gpu_harness import Lockstep
    from oracle import OracleBoard
    B, n = 3, 40
    e = _engine(B, n, seed=2)
    boards = [OracleBoard() for _ in range(B)]
    ls = Lockstep(e, boards, kind="hash_sharp", salts=[5, 6, 7])
    # a repetition cycle on board 0 and 1 (knights out and back, twice), something else on board 2
    cyc = (["b0c2", "b9c7", "c2b0", "c7b9"] * 3)[:11]   # after 11 plies the reply c7b9 repeats the start position a 4th time
    other = ["h2e2", "h9g7", "e2e6", "g7e6", "b0c2", "b9c7", "a0a1", "a9a8", "i0i1", "i9i8", "a1a2"]
    import oracle
    L = oracle.lib()
    uid = lambda u: L.xq_move_id((ord(u[0]) - 97) + 9 * int(u[1]), (ord(u[2]) - 97) + 9 * int(u[3]))
    for i in range(11):
        assert all(uid(u) in boards[b].legal_ids() for b, u in enumerate((cyc[i], cyc[i], other[i])))
        ls.play([uid(cyc[i]), uid(cyc[i]), uid(other[i])])
    for _ in range(n):
        ls.step(check_leaf=True)
    ls.compare_roots()
    plies0, pos0 = e.game_status()["plies"].copy(), e.root_positions().copy()
    e.reset_tree(np.array([1, 0, 1], np.uint8))
    ls.mcts[0].update_with_move(-1)
    ls.mcts[2].update_with_move(-1)
    rc = e.root_children()
    assert rc["k"][0] == 0 and rc["k"][2] == 0 and rc["root_visits"][0] == 0 and rc["k"][1] > 0
    assert np.array_equal(e.game_status()["plies"], plies0) and np.array_equal(e.root_positions(), pos0)
    # the search after the reset still sees the game history: the start position is on the chain three times, so the
    # reply that repeats it a fourth time is a draw leaf for engine and oracle alike (leaf status compared every step)
    t0 = e.stats()["terminal_leaves"]
    for _ in range(n):
        ls.step(check_leaf=True)
    ls.compare_roots()
    assert e.stats()["terminal_leaves"] > t0
    e.check_healthy()


def test_rule_flags_are_validated_and_error_bits():
    from chinesechesszero_amd import _lib
    from chinesechesszero_amd._lib import CczError
    L = _lib.lib()
    cfg = _lib.Config(n_boards=2, n_playout=4, c_puct=5, eps=0.25, alpha=0.2, temp=1.0, rule_flags=_lib.RULE_PERPETUAL_CHECK)
    h = C.c_void_p()
    assert L.ccz_create(C.byref(cfg), C.byref(h)) == 0      # round 2 refused the flag (-6, "not implemented"); it is a rule now
    assert L.ccz_destroy(h) == 0
    cfg.rule_flags = 8
    assert L.ccz_create(C.byref(cfg), C.byref(h)) == -1 and b"unknown rule_flags" in L.ccz_last_error()
    # the two overflow kinds report separate bits (round 1 shared bit 2)
    assert _lib.ERR_BITS[2] != _lib.ERR_BITS[64] and "chain" in _lib.ERR_BITS[64] and "depth" in _lib.ERR_BITS[2]


def test_float16_value_quirk_matches_oracle_and_differs_from_float32():
    """CCZ_FLAG_VALUE_F16: Q accumulated in float16 as on the reference's CUDA (autocast) path -- lockstep against the
    oracle's float16 twin (itself pinned by two golden traces of the reference run with a float16 value), fused and
    two-kernel launch sequences, terminal leaves included; and NOT what the default float32 engine computes."""
    from gpu_harness import Lockstep
    from golden_cases import STARTS
    from oracle import OracleBoard
    B, n = 6, 120
    e = _engine(B, n, seed=5, value_f16=True)
    e32 = _engine(B, n, seed=5)
    boards, boards32 = [], []
    for b in range(B):
        if b % 3 == 2:   # an endgame with mates in reach: terminal (Python-float) values meet float16 nodes
            e.set_position(b, STARTS["two_rooks"], 1, 0)
            e32.set_position(b, STARTS["two_rooks"], 1, 0)
            boards.append(OracleBoard.from_array(STARTS["two_rooks"], 1, 0))
            boards32.append(OracleBoard.from_array(STARTS["two_rooks"], 1, 0))
        else:
            boards.append(OracleBoard())
            boards32.append(OracleBoard())
    salts = [41, 42, 43, 44, 45, 46]
    ls = Lockstep(e, boards, kind="hash_sharp", salts=salts, value_f16=True)
    ls32 = Lockstep(e32, boards32, kind="hash_sharp", salts=salts)
    for ply in range(2):
        if ply == 0:
            ls.run_fused(n, check_leaf=True)
            ls32.run_fused(n, check_leaf=False)
        else:
            for _ in range(n):
                ls.step(check_leaf=True)
                ls32.step(check_leaf=False)
        rc, rc32 = ls.compare_roots(), ls32.compare_roots()
        q = np.concatenate([rc["q"][b][:rc["k"][b]] for b in range(B)])
        assert np.array_equal(q, q.astype(np.float16).astype(np.float32))      # every Q is a float16 number
        assert not all(np.array_equal(rc["q"][b], rc32["q"][b]) for b in range(B))
        pick = lambda r, b: int(r["acts"][b][int(np.argmax(r["visits"][b][:r["k"][b]]))]) if r["k"][b] else -1  # -1: game over
        mv = [pick(rc, b) for b in range(B)]
        mv32 = [pick(rc32, b) for b in range(B)]
        ls.play(mv)
        ls32.play(mv32)
    assert e.stats()["terminal_leaves"] > 0
    e.check_healthy()


def test_type_major_scan_order_matches_oracle_and_reaches_the_host_mirror(rules_of_case):
    """A POSITION-DEPENDENT `legal_moves` order -- major key = piece type of the mover (ccz_config.type_rank), minor key =
    move_rank: the python-chess-family scheme "non-pawn moves by from / to square descending, then pawn moves". A static
    permutation of the ids cannot express it (the same id is a pawn move in one position and a rook move in another)."""
    from gpu_harness import Lockstep
    from golden_cases import STARTS
    from oracle import OracleBoard
    from chinesechesszero_amd.game import Board
    from chinesechesszero_amd.mcts import MCTS
    rank, trank = rules_of_case({"order": "scan_desc_pawns_last"}, product=True, both=True)
    B, n = 4, 70
    e = _engine(B, n, seed=3, move_rank=rank, type_rank=trank)
    boards = []
    for b in range(B):
        if b == 3:
            e.set_position(b, STARTS["wide80"], 1, 0)       # 80 legal moves: both 64-lane halves of the partition
            boards.append(OracleBoard.from_array(STARTS["wide80"], 1, 0))
        else:
            boards.append(OracleBoard())
    ls = Lockstep(e, boards, kind="hash_sharp", salts=[51, 52, 53, 54])
    for ply in range(2):
        for _ in range(n):
            ls.step(check_leaf=True)           # leaf id lists vs oracle.legal_ids() in the installed order, every step
        rc = ls.compare_roots()
        ls.play([int(rc["acts"][b][int(np.argmax(rc["visits"][b][:rc["k"][b]]))]) for b in range(B)])
    e.check_healthy()
    # pawn moves come last at the opening position: the five pawn pushes close the list
    ob = OracleBoard()
    ids = ob.legal_ids()
    import oracle
    L = oracle.lib()
    is_pawn = [(ob.squares()[L.xq_move_from(i)] & 7) == 1 for i in ids]
    assert is_pawn == [False] * 39 + [True] * 5
    # host mirror: Board and the engines MCTS creates follow tools.set_rules
    hb = Board()
    assert hb.legal_ids() == ids
    m = MCTS(lambda board, r=None, b=None: (zip(board.legal_ids(), np.full(len(board.legal_ids()), 1.0 / 2086, np.float32)), np.zeros((1, 1), np.float32)),
             c_puct=5, n_playout=10)
    acts, _ = m.get_move_probs(hb, temp=1.0)
    assert list(acts) == ids and m._engine.type_rank == tuple(trank)
    with pytest.raises(Exception, match="type_rank"):
        _engine(1, 4, type_rank=[0, 9, 0, 0, 0, 0, 0, 0])


def test_pawn_move_clock_rule_is_switchable():
    """CCZ_RULE_PAWN_MOVE_RESETS_CLOCK: the sixty-move clock (and the repetition history) restarts on pawn moves too --
    python-chess's `is_zeroing`, which the cchess port may have kept. Engine, oracle and host Board flip together."""
    import oracle
    from gpu_harness import Lockstep
    from golden_cases import STARTS, sq as S
    from oracle import OracleBoard
    from chinesechesszero_amd import tools
    from chinesechesszero_amd.game import Board
    L = oracle.lib()
    uid = lambda u: L.xq_move_id(S(u[:2]), S(u[2:]))
    pos = STARTS["pawns"]                      # kings and five pawns each; RED to move, 118 plies without a capture
    seq = ["a3a4", "f9f8", "d0d1"]            # pawn move, king move, king move
    results = {}
    for flag in (False, True):
        try:
            oracle.set_rules(pawn_move_resets_clock=flag)
            tools.set_rules(pawn_move_resets_clock=flag)
            e = _engine(1, 40, seed=1)        # picks the rule up from tools.set_rules
            assert e.pawn_move_resets_clock is flag
            e.set_position(0, pos, 1, 118)
            ob = OracleBoard.from_array(pos, 1, 118)
            hb = Board(pos, True, 118)
            over = []
            for u in seq:
                e.finish_move(forced_moves=np.array([uid(u)], np.int32), keep_tree=False)
                ob.push(u)
                hb.push(u)
                st = e.game_status()
                assert bool(st["over"][0]) == ob.is_game_over() == hb.is_game_over(), (flag, u)
                assert hb.halfmove_clock == ob.halfmove
                over.append(bool(st["over"][0]))
                if st["over"][0]:
                    break
            results[flag] = over
            if not over[-1]:                   # searches below the new root agree on clocks and draws leaf by leaf
                ls = Lockstep(e, [ob], kind="hash_sharp", salts=[9])
                ls.run_fused(40, check_leaf=True)
                ls.compare_roots()
            e.check_healthy()
        finally:
            oracle.set_rules()
            tools.set_rules()
    assert results[False] == [False, True]     # 118 + 2 quiet plies: the sixty-move rule ends the game
    assert results[True] == [False, False, False]   # the pawn move restarted the clock


def _perpetual_case(checker_is_red: bool):
    """A rook that checks a bare king back and forth (a9+ Ke8, a8+ Ke9, ...): (squares, side to move, the 4-ply cycle)."""
    from golden_cases import sq as S
    pos = np.zeros(90, np.uint8)
    if checker_is_red:
        pos[S("d0")] = 7            # red king
        pos[S("a8")] = 3            # red rook
        pos[S("e9")] = 7 + 8        # black king
        return pos, 1, ["a8a9", "e9e8", "a9a8", "e8e9"]
    pos[S("d9")] = 7 + 8
    pos[S("a1")] = 3 + 8            # black rook
    pos[S("e0")] = 7
    return pos, 0, ["a1a0", "e0e1", "a0a1", "e1e0"]


@pytest.mark.parametrize("checker_is_red", [True, False])
def test_perpetual_check_rule_decides_a_fourfold_repetition(checker_is_red):
    """CCZ_RULE_PERPETUAL_CHECK: the side that checked with every move of a repetition cycle loses.
    Engine (k_finish_move), oracle (xq_outcome_winner) and host Board flip together; without the flag the same game is a draw;
    a repetition without checks stays a draw under the flag; the search below such a root is unchanged (leaf value 0.0)."""
    import oracle
    from gpu_harness import Lockstep
    from golden_cases import sq as S
    from oracle import OracleBoard
    from chinesechesszero_amd import tools
    from chinesechesszero_amd.game import Board
    L = oracle.lib()
    uid = lambda u: L.xq_move_id(S(u[:2]), S(u[2:]))
    pos, turn, cycle = _perpetual_case(checker_is_red)
    quiet = ["a8a7", "e9f9", "a7a8", "f9e9"] if checker_is_red else ["a1a2", "e0f0", "a2a1", "f0e0"]   # the rook shuffles, no check
    results = {}
    for flag, seq, name in ((True, cycle, "perpetual"), (False, cycle, "flag off"), (True, quiet, "no checks")):
        try:
            oracle.set_rules(perpetual_check=flag)
            tools.set_rules(perpetual_check=flag)
            e = _engine(1, 24, seed=3)          # picks the rule up from tools.set_rules
            assert e.perpetual_check is flag
            e.set_position(0, pos, turn, 0)
            ob, hb = OracleBoard.from_array(pos, turn, 0), Board(pos, bool(turn), 0)
            for ply in range(12):
                if ply == 8:
                    # two cycles played, the position has occurred three times: a search from here meets the fourth
                    # occurrence as a LEAF -- value 0.0 with or without the flag (mcts.py:120-122 `end and is_tie`), visit
                    # counts bit-exact with the oracle's sequential search
                    ls = Lockstep(e, [ob], kind="hash_sharp", salts=[5])
                    ls.run_fused(24, check_leaf=True)
                    ls.compare_roots()
                u = seq[ply % 4]
                e.finish_move(forced_moves=np.array([uid(u)], np.int32), keep_tree=False)
                ob.push(u)
                hb.push(u)
                st = e.game_status()
                assert bool(st["over"][0]) == ob.is_game_over() == hb.is_game_over() == (ply == 11), (name, ply)
            w = int(st["winner"][0])
            o, h = ob.outcome(), hb.outcome()
            ow = -1 if o.winner is None else int(o.winner)
            hw = -1 if h.winner is None else int(h.winner)
            assert w == ow == hw, (name, w, ow, hw)
            assert h.termination == ("perpetual_check" if w >= 0 else "fourfold_repetition")
            results[name] = w
            # z of the harvested tuples follows the adjudicated winner (game.py:213-219)
            _, _, z = e.harvest()
            zz = z.cpu().numpy()[:12]
            want = [0.0] * 12 if w < 0 else [1.0 if ((turn if t % 2 == 0 else 1 - turn) == w) else -1.0 for t in range(12)]
            assert zz.tolist() == want, (name, zz.tolist())
            e.check_healthy()
        finally:
            oracle.set_rules()
            tools.set_rules()
    loser_is_red = checker_is_red
    assert results == {"perpetual": 0 if loser_is_red else 1, "flag off": -1, "no checks": -1}


def test_perpetual_check_that_coincides_with_the_sixty_move_draw_is_a_draw():
    """with CCZ_RULE_PERPETUAL_CHECK a repetition that completes at the very ply the sixty-move clock reaches 120 is a
    DRAW -- the order of the reference's checks (game.py:208-214: insufficient material, sixty moves, then repetition), which the
    host Board always followed; k_finish_move and the oracle adjudicated a perpetual-check winner there. z follows."""
    import oracle
    from golden_cases import sq as S
    from oracle import OracleBoard
    from chinesechesszero_amd import tools
    from chinesechesszero_amd.game import Board
    L = oracle.lib()
    uid = lambda u: L.xq_move_id(S(u[:2]), S(u[2:]))
    pos, turn, cycle = _perpetual_case(True)
    try:
        oracle.set_rules(perpetual_check=True)
        tools.set_rules(perpetual_check=True)
        for clock0, want_winner in ((108, -1), (100, 0)):     # 12 plies later the clock stands at 120 (draw first) / at 112 (red, the checker, loses)
            e = _engine(1, 8, seed=3)
            e.set_position(0, pos, turn, clock0)
            ob, hb = OracleBoard.from_array(pos, turn, clock0), Board(pos, bool(turn), clock0)
            for ply in range(12):
                u = cycle[ply % 4]
                e.finish_move(forced_moves=np.array([uid(u)], np.int32), keep_tree=False)
                ob.push(u)
                hb.push(u)
            st = e.game_status()
            assert st["over"][0] and ob.is_game_over() and hb.is_game_over()
            o, h = ob.outcome(), hb.outcome()
            got = (int(st["winner"][0]), -1 if o.winner is None else int(o.winner), -1 if h.winner is None else int(h.winner))
            assert got == (want_winner,) * 3, (clock0, got)
            _, _, z = e.harvest()
            assert (float(z.abs().max()) == 0.0) == (want_winner < 0)
            e.check_healthy()
    finally:
        oracle.set_rules()
        tools.set_rules()


def test_probe_cchess_round_trip_on_the_products_own_board():
    """tools/probe_cchess.py run against the PRODUCT's host Board (rules answered by the HIP movegen kernel) duck-typed as
    `cchess`: it reads back the canonical preset with nothing unsupported, and its golden file replays on the oracle -- i.e. the
    kernels and the oracle give the probe the same answers on its ~200 positions and crafted endings."""
    import importlib.util
    import os
    import oracle
    from fake_cchess import make_module
    from test_oracle_rules import _replay_golden_on_oracle
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("probe_cchess", os.path.join(root, "tools", "probe_cchess.py"))
    P = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(P)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        got, golden = P.probe(make_module("host"), d, n_games=6, plies=26)
        assert got["unsupported_differences"] == [] and got["move_rank"] is None and got["type_rank"] is None
        assert tuple(got["plane_of_type"]) == (0, 0, 1, 2, 3, 4, 5, 6) and not got["perpetual_check"] and not got["pawn_move_resets_clock"]
        oracle.set_rules()
        assert _replay_golden_on_oracle(os.path.join(d, "cchess_golden.npz")) == len(golden) > 140


def test_rule_presets_install_tables_in_every_layer():
    """tools.set_rules(preset=...): "canonical" and the unverified "python-chess-lineage" guess (planes P,R,N,B,A,K,C; piece-set
    scan order with pawns last; perpetual check). The engine, the host Board and -- given the same tables -- the oracle agree
    on move order and planes leaf by leaf under the preset."""
    import oracle
    from gpu_harness import Lockstep
    from oracle import OracleBoard
    from chinesechesszero_amd import tools
    from chinesechesszero_amd.game import Board
    try:
        with pytest.raises(ValueError):
            tools.set_rules(preset="no such preset")
        tools.set_rules(preset="python-chess-lineage")
        assert tools.PRESET == "python-chess-lineage" and tools.PERPETUAL_CHECK and tools.PLANE_OF_TYPE == (0, 0, 6, 1, 2, 3, 4, 5)
        oracle.set_rules(**tools.current_rules())
        e = _engine(2, 48, seed=2)
        assert e.perpetual_check and e.plane_of_type == (0, 0, 6, 1, 2, 3, 4, 5)
        ob = [OracleBoard(), OracleBoard()]
        hb = Board()
        ids = hb.legal_ids()
        assert ids == ob[0].legal_ids() and len(ids) == 44
        fr = [int(tools.MOVE_FROM[i]) for i in ids]
        pawn = [(int(hb.squares()[f]) & 7) == 1 for f in fr]
        assert pawn == sorted(pawn) and sum(pawn) == 5                          # the five pawn moves come last
        nonp = [f for f, p in zip(fr, pawn) if not p]
        assert nonp == sorted(nonp, reverse=True)                              # from-squares in descending scan order
        red, black = tools.decode_board(hb)
        assert red[0].sum() == 5 and red[6].sum() == 2 and red[1].sum() == 2 and red[5].sum() == 1   # P, C, R, K planes
        ls = Lockstep(e, ob, kind="hash_sharp", salts=[1, 2])
        ls.run_fused(48, check_leaf=True)                                      # leaf ids AND planes vs the oracle every step
        ls.compare_roots()
        e.check_healthy()
        tools.set_rules(preset="canonical")
        assert tools.PRESET == "canonical" and tools.MOVE_RANK is None and not tools.PERPETUAL_CHECK
        assert Board().legal_ids() == sorted(Board().legal_ids())
        tools.set_rules(preset="python-chess-lineage", perpetual_check=False, plane_of_type=(0, 0, 1, 2, 3, 4, 5, 6))
        assert tools.PRESET == "custom" and tools.MOVE_RANK is not None
    finally:
        oracle.set_rules()
        tools.set_rules()


@pytest.mark.parametrize("with_move_rank", [False, True])
def test_random_eight_class_type_rank_on_positions_with_more_than_64_legal_moves(with_move_rank):
    """the stable partition by the mover's piece type (two ballots per class) on lists longer than one wave --
    the second-half path (entries 64..127) and several classes at once -- against the oracle's comparison sort."""
    import oracle
    from golden_cases import STARTS
    from oracle import OracleBoard
    rs = np.random.RandomState(77 + int(with_move_rank))
    positions = [(STARTS["wide80"], 1, 0), (STARTS["wide80"], 0, 0), (None, 1, 0)]
    for trial in range(4):
        type_rank = [0] + rs.randint(0, 8, size=7).tolist()            # up to 8 classes, ties between types included
        rank = rs.permutation(2086).astype(np.uint16) if with_move_rank else None
        try:
            oracle.set_rules(move_rank=rank, type_rank=type_rank)
            e = _engine(len(positions), 8, seed=trial, move_rank=rank, type_rank=type_rank)
            boards = []
            for b, (sq, turn, half) in enumerate(positions):
                ob = OracleBoard() if sq is None else OracleBoard.from_array(sq, turn, half)
                if sq is not None:
                    e.set_position(b, sq, turn, half)
                boards.append(ob)
            e.select_leaves()
            info = e.leaf_info()
            ks = []
            for b, ob in enumerate(boards):
                want = ob.legal_ids()
                k = int(info["k"][b])
                ks.append(k)
                assert info["ids"][b][:k].tolist() == want, (trial, b, type_rank)
            assert max(ks) > 64
            e.check_healthy()
        finally:
            oracle.set_rules()
