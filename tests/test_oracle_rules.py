"""CPU: rules of the oracle -- published perft counts and internal consistency properties.

Rules parity with the reference's third-party `cchess` is unpinned (source absent); these tests
anchor the restatement to the published Xiangqi start-position perft and to properties.
"""
import numpy as np
import pytest

import oracle
from oracle import OracleBoard

PERFT = {1: 44, 2: 1920, 3: 79666, 4: 3290240}


@pytest.mark.parametrize("depth", [1, 2, 3, 4])
def test_perft_start_position(depth):
    assert OracleBoard().perft(depth) == PERFT[depth]


@pytest.mark.slow
def test_perft5_start_position():
    assert OracleBoard().perft(5) == 133312995


def random_walk_positions(n_games, max_plies, seed):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n_games):
        b = OracleBoard()
        for _ in range(max_plies):
            ids = b.legal_ids()
            if not ids or b.is_game_over():
                break
            out.append(b.copy())
            b.push_id(ids[rs.randint(len(ids))])
    return out


def test_in_check_fast_equals_slow_on_random_positions():
    L = oracle.lib()
    import ctypes as C
    for b in random_walk_positions(30, 150, 1):
        for color in (0, 1):
            assert L.xq_in_check(C.byref(b.b.pos), color) == L.xq_in_check_slow(C.byref(b.b.pos), color)
        # every pseudo-legal move: legality judged identically by both attack tests
        fr = (C.c_uint8 * 256)()
        to = (C.c_uint8 * 256)()
        col = int(b.b.pos.turn)
        n = L.xq_pseudo_moves(C.byref(b.b.pos), col, fr, to)
        for i in range(0, n, 3):
            q = oracle.Pos()
            C.memmove(C.byref(q), C.byref(b.b.pos), C.sizeof(oracle.Pos))
            q.sq[to[i]] = q.sq[fr[i]]
            q.sq[fr[i]] = 0
            assert L.xq_in_check(C.byref(q), col) == L.xq_in_check_slow(C.byref(q), col)


def test_legal_ids_sorted_unique_and_in_table():
    for b in random_walk_positions(10, 120, 2):
        ids = b.legal_ids()
        assert ids == sorted(set(ids))
        assert all(0 <= i < 2086 for i in ids)


def test_mirror_symmetry_of_legal_moves():
    """legal(mirror(s)) == mirror(legal(s)) through the reference's flip map."""
    fm = oracle.flip_map()
    for b in random_walk_positions(10, 100, 3):
        sq = b.squares().reshape(10, 9)[:, ::-1].reshape(-1)
        m = OracleBoard.from_array(sq, int(b.turn), b.halfmove)
        assert sorted(fm[b.legal_ids()].tolist()) == m.legal_ids()


def test_colour_symmetry_of_legal_move_count():
    """Rotating the board by 180 degrees and swapping colours preserves the number of legal moves."""
    for b in random_walk_positions(10, 100, 4):
        sq = b.squares()[::-1].copy()
        nz = sq != 0
        sq[nz] = sq[nz] ^ 8
        m = OracleBoard.from_array(sq, 0 if b.turn else 1, b.halfmove)
        assert len(m.legal_ids()) == len(b.legal_ids())


def test_repetition_and_sixty_move_predicates():
    b = OracleBoard()
    # shuffle knights back and forth: b0c2 b9c7 c2b0 c7b9 repeated
    cyc = ["b0c2", "b9c7", "c2b0", "c7b9"]
    assert not b.is_fourfold_repetition()
    for rep in range(3):
        for m in cyc:
            b.push(m)
    # the start position has now occurred 4 times (initial + 3 returns)
    assert b.is_fourfold_repetition() and b.is_tie() and b.is_game_over()
    assert b.outcome().winner is None
    c = OracleBoard.from_array(b.squares(), 1, 119)
    assert not c.is_sixty_moves()
    c.push("b0c2")
    assert c.halfmove == 120 and c.is_sixty_moves() and c.is_game_over()
    d = OracleBoard.from_array(b.squares(), 1, 119)
    d.push("b2b9")  # cannon captures the knight: clock resets
    assert d.halfmove == 0 and not d.is_sixty_moves()


def test_insufficient_material_and_no_legal_moves():
    from golden_cases import STARTS
    b = OracleBoard.from_array(STARTS["capture_to_bare"], 1, 0)
    assert not b.is_insufficient_material()
    b.push("e0e1")
    assert b.is_insufficient_material() and b.is_game_over() and b.outcome().winner is None
    # two rooks mate: a7a9 mates the bare king
    m = OracleBoard.from_array(STARTS["two_rooks"], 1, 0)
    m.push("a7a9")
    assert m.legal_ids() == [] and m.is_game_over() and not m.is_tie()
    assert m.outcome().winner is True  # RED wins: side to move (BLACK) has no legal move


def test_leaf_planes_layout():
    b = OracleBoard()
    p = b.leaf_planes()
    red, black = b.decode()
    assert p.shape == (17, 7, 10, 9)
    assert np.array_equal(p[7], red) and np.array_equal(p[15], black)
    assert np.all(p[16] == 1) and p[:7].sum() == 0 and p[8:15].sum() == 0
    # channel = piece_type - 1 with PAWN=1..KING=7; red king on e0, black king on e9
    assert red[6, 0, 4] == 1 and black[6, 9, 4] == 1 and red[0, 3, 0] == 1
    b.push("b0c2")
    assert np.all(b.leaf_planes()[16] == 0)


def test_perpetual_check_rule_of_the_oracle():
    """xq_set_perpetual_check (twin of CCZ_RULE_PERPETUAL_CHECK): in a fourfold repetition the side that
    checked with every move of the cycle loses; off by default; a repetition without checks stays a draw; is_tie / is_game_over
    (what the search consults, mcts.py:116-117) do not depend on the flag."""
    import numpy as np

    import oracle
    from golden_cases import sq as S
    from oracle import OracleBoard
    pos = np.zeros(90, np.uint8)
    pos[S("d0")], pos[S("a8")], pos[S("e9")] = 7, 3, 15
    cycle, quiet = ["a8a9", "e9e8", "a9a8", "e8e9"], ["a8a7", "e9f9", "a7a8", "f9e9"]
    got = {}
    try:
        for flag, seq, name in ((True, cycle, "perpetual"), (False, cycle, "off"), (True, quiet, "quiet")):
            oracle.set_rules(perpetual_check=flag)
            b = OracleBoard.from_array(pos, 1, 0)
            for ply in range(12):
                assert seq[ply % 4] in b.legal_moves
                b.push(seq[ply % 4])
                assert b.is_game_over() == b.is_tie() == (ply == 11)
            assert b.is_fourfold_repetition()
            got[name] = b.outcome().winner
    finally:
        oracle.set_rules()
    assert got == {"perpetual": False, "off": None, "quiet": None}     # RED checked throughout: BLACK wins
