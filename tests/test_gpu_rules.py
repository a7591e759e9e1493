"""GPU: move generation / game-end flags of the HIP engine against the CPU oracle (bit-exact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bfs_positions(depth):
    from oracle import OracleBoard
    level = [OracleBoard()]
    out = [level]
    for _ in range(depth):
        nxt = []
        for b in level:
            for i in b.legal_ids():
                c = b.copy()
                c.push_id(i)
                nxt.append(c)
        out.append(nxt)
        level = nxt
    return out


def _compare(boards, chunk=20000):
    from chinesechesszero_amd.engine import legal_moves
    total = 0
    for s in range(0, len(boards), chunk):
        bs = boards[s:s + chunk]
        sq = np.stack([b.squares() for b in bs])
        turn = np.array([1 if b.turn else 0 for b in bs], np.uint8)
        half = np.array([b.halfmove for b in bs], np.int32)
        mask, cnt, flags = legal_moves(sq, turn, half)
        for j, b in enumerate(bs):
            ids = b.legal_ids()
            got = np.nonzero(mask[j])[0].tolist()
            assert got == ids, (s + j, got, ids)
            assert cnt[j] == len(ids)
            assert bool(flags[j] & 1) == b.in_check()
            assert bool(flags[j] & 2) == b.is_insufficient_material()
            assert bool(flags[j] & 4) == b.is_sixty_moves()
            assert not (flags[j] & 128)
        total += int(cnt.sum())
    return total


def test_perft_positions_match_oracle_and_published_counts():
    levels = _bfs_positions(3)  # 1 + 44 + 1920 + 79666 positions
    assert [len(l) for l in levels] == [1, 44, 1920, 79666]
    assert _compare(levels[0]) == 44
    assert _compare(levels[1]) == 1920
    assert _compare(levels[2]) == 79666
    assert _compare(levels[3]) == 3290240  # published perft(4)


def test_random_midgame_and_endgame_positions():
    from oracle import OracleBoard
    rs = np.random.RandomState(7)
    boards = []
    for _ in range(60):
        b = OracleBoard()
        for _ in range(400):
            ids = b.legal_ids()
            if not ids or b.is_game_over():
                break
            boards.append(b.copy())
            b.push_id(ids[rs.randint(len(ids))])
        boards.append(b.copy())  # final (possibly terminal) position
    assert len(boards) > 5000
    _compare(boards)


def test_golden_endgames_and_mirror_symmetry():
    from golden_cases import STARTS
    from oracle import OracleBoard, flip_map
    from chinesechesszero_amd.engine import legal_moves
    fm = flip_map()
    boards = [OracleBoard.from_array(sq, t, h) for sq in STARTS.values() for t in (0, 1) for h in (0, 119, 120)]
    _compare(boards)
    sq = np.stack([b.squares() for b in boards])
    turn = np.array([1 if b.turn else 0 for b in boards], np.uint8)
    mask, _, _ = legal_moves(sq, turn)
    msq = sq.reshape(-1, 10, 9)[:, :, ::-1].reshape(-1, 90)
    mmask, _, _ = legal_moves(msq, turn)
    assert np.array_equal(mmask, mask[:, fm])


def test_apply_moves_matches_oracle_push():
    from oracle import OracleBoard
    from chinesechesszero_amd.engine import apply_moves
    rs = np.random.RandomState(3)
    boards, moves = [], []
    b = OracleBoard()
    for _ in range(300):
        ids = b.legal_ids()
        if not ids or b.is_game_over():
            b = OracleBoard()
            ids = b.legal_ids()
        m = ids[rs.randint(len(ids))]
        boards.append(b.copy())
        moves.append(m)
        b.push_id(m)
    sq = np.stack([x.squares() for x in boards])
    turn = np.array([1 if x.turn else 0 for x in boards], np.uint8)
    nsq, nturn, cap = apply_moves(sq, turn, np.array(moves, np.int32))
    for j, x in enumerate(boards):
        was = x.squares()[oracle_to(moves[j])]
        x.push_id(moves[j])
        assert np.array_equal(nsq[j], x.squares())
        assert nturn[j] == (1 if x.turn else 0)
        assert cap[j] == was


def oracle_to(mid):
    import oracle
    return oracle.lib().xq_move_to(int(mid))


def test_action_table_and_flip_map_from_library(golden):
    from chinesechesszero_amd import tools
    assert [tools.move_id2move_action[i] for i in range(2086)] == golden["table"]
    assert np.array_equal(tools.flip_map(), golden["data"]["flip_map"])
