"""CPU: the C oracle against the golden vectors produced by executing the reference's own code."""
import hashlib

import numpy as np
import pytest

import oracle
from oracle import OracleBoard, OracleMCTS
from oracle.evaluators import EVALUATORS

from golden_cases import case_start


def test_action_table_matches_reference(golden):
    t = oracle.move_table()
    assert t == golden["table"]
    assert hashlib.sha256(",".join(t).encode()).hexdigest() == golden["meta"]["table_sha256"]
    L = oracle.lib()
    for i, s in enumerate(t):
        fr, to = L.xq_move_from(i), L.xq_move_to(i)
        assert s == "abcdefghi"[fr % 9] + str(fr // 9) + "abcdefghi"[to % 9] + str(to // 9)
        assert L.xq_move_id(fr, to) == i


def test_flip_map_matches_reference(golden):
    fm = oracle.flip_map()
    assert np.array_equal(fm, golden["data"]["flip_map"])
    assert np.array_equal(fm[fm], np.arange(2086))  # involution
    assert int((fm == np.arange(2086)).sum()) == 90


def test_dtype_facts(golden):
    f = golden["meta"]["dtype_facts"]
    # the promotion rules the oracle's V_F32 / V_PYFLOAT state machine restates (mcts.py:41-78)
    assert f["q_dtype_after_net_backup"] == "float32"
    assert f["puct_dtype"] == "float64"
    assert f["cpuct_times_prob_dtype"] == "float32"
    assert f["sqrt_int_dtype"] == "float64"
    assert f["q_type_terminal_only"] == "float"
    assert f["q_dtype_mixed"] == "float32"
    assert f["unvisited_is_inf"] is True
    # the CUDA (autocast) path of the reference hands Node.update a float16 value: Q is then float16, PUCT still float64
    assert f["q_dtype_after_f16_backup"] == "float16" and f["puct_dtype_with_f16_q"] == "float64"
    assert f["q_dtype_f16_then_terminal"] == "float16"


def test_pi_from_visits(golden):
    d = golden["data"]
    L = oracle.lib()
    for i in range(golden["meta"]["n_pi"]):
        visits, temp, ref = d[f"pi{i}_visits"], float(d[f"pi{i}_temp"]), d[f"pi{i}_pi"]
        # libm restatement of softmax(1/temp*log(N+1e-10))
        x = 1.0 / temp * np.log(visits.astype(np.int64) + 1e-10)
        e = np.exp(x - x.max())
        assert np.allclose(e / e.sum(), ref, rtol=0, atol=1e-15)
        # deterministic (device-mode) twin stays within 1e-12 of the reference
        assert np.allclose(oracle.det_pi(visits, temp), ref, rtol=0, atol=1e-12)


def _make_eval(name):
    ev = EVALUATORS[name]

    def f(board, ids):
        p, v = ev(board.squares()[None, :], np.array([1 if board.turn else 0]))
        return p[0][ids], v[0]

    return f


N_CASES = 20  # 13 canonical + 2 shuffled order (order_seed) + 2 float16 value + PLAYOUT = 1600 + 2 type-major scan order


def test_case_count(golden):
    assert len(golden["meta"]["cases"]) == N_CASES
    assert sum("order_seed" in c for c in golden["meta"]["cases"]) == 2
    assert sum(c.get("value_dtype") == "float16" for c in golden["meta"]["cases"]) == 2
    assert sum(c.get("order") == "scan_desc_pawns_last" for c in golden["meta"]["cases"]) == 2


@pytest.mark.parametrize("idx", range(N_CASES))
def test_search_trace_matches_reference(golden, idx, rules_of_case):
    """Visit counts, Q, priors bit-exact; pi to 1e-12; sampled moves identical under np.random.seed."""
    case = golden["meta"]["cases"][idx]
    rules_of_case(case)
    d = golden["data"]
    name = case["name"]
    sqs, turn, half = case_start(case)
    board = OracleBoard.from_array(sqs, turn, half) if case["start"] != "start" else OracleBoard()
    mcts = OracleMCTS(_make_eval(case["ev"]), c_puct=5, n_playout=case["n"], value_f16=case.get("value_dtype") == "float16")
    rs = np.random.RandomState(case["seed"])
    for ply in range(case["plies_done"]):
        acts, visits, probs = mcts.get_move_probs(board, case["temps"][ply])
        a2, v2, q, prior = mcts.root_children()
        assert np.array_equal(acts, d[f"{name}_p{ply}_acts"]), (name, ply)
        assert np.array_equal(visits, d[f"{name}_p{ply}_visits"]), (name, ply)
        assert np.array_equal(q.view(np.uint32), d[f"{name}_p{ply}_q"].view(np.uint32)), (name, ply)
        assert np.array_equal(prior.view(np.uint32), d[f"{name}_p{ply}_prior"].view(np.uint32))
        assert mcts.root_visits() == int(d[f"{name}_p{ply}_rootvisits"])
        assert np.allclose(probs, d[f"{name}_p{ply}_pi"], rtol=0, atol=1e-12)
        # reference-exact sampling: mcts.py:216-229 on the golden pi
        pi = d[f"{name}_p{ply}_pi"]
        if case["selfplay"]:
            mixed = 0.75 * pi + 0.25 * rs.dirichlet(0.2 * np.ones(len(pi)))
            assert np.array_equal(mixed, d[f"{name}_p{ply}_mixed"])
            move = int(rs.choice(acts, p=mixed))
            mcts.update_with_move(move)
        else:
            move = int(rs.choice(acts, p=pi))
            mcts.update_with_move(-1)
        assert move == int(d[f"{name}_p{ply}_move"])
        board.push_id(move)
    assert np.array_equal(board.squares(), d[f"{name}_final_sq"])


def test_terminal_values_reached_in_traces(golden):
    """The endgame cases must actually exercise terminal leaves (fewer evaluator calls than playouts)."""
    ev_hit = {c["name"]: c["evals"] for c in golden["meta"]["cases"]}
    # the reference calls the evaluator on every playout incl. terminal leaves
    for c in golden["meta"]["cases"]:
        assert ev_hit[c["name"]] == c["n"] * c["plies_done"]
    case = [c for c in golden["meta"]["cases"] if c["name"] == "rooks_sixty_n300"][0]
    sqs, turn, half = case_start(case)
    board = OracleBoard.from_array(sqs, turn, half)
    mcts = OracleMCTS(_make_eval(case["ev"]), c_puct=5, n_playout=case["n"])
    mcts.get_move_probs(board, 1.0)
    assert mcts.n_evals < case["n"]  # the oracle skips the evaluator on terminal leaves (results-neutral)


def test_shuffled_order_changes_the_search(golden):
    """The shuffled-order traces are NOT what the canonical order gives: the order is load-bearing (mcts.py:47-48,59-61)."""
    case = [c for c in golden["meta"]["cases"] if c["name"] == "start_sharp_shuffled_n200"][0]
    d = golden["data"]
    mcts = OracleMCTS(_make_eval(case["ev"]), c_puct=5, n_playout=case["n"])
    acts, visits, _ = mcts.get_move_probs(OracleBoard(), 1.0)
    assert sorted(acts.tolist()) == sorted(d["start_sharp_shuffled_n200_p0_acts"].tolist())
    assert not np.array_equal(acts, d["start_sharp_shuffled_n200_p0_acts"])
    by_id = dict(zip(d["start_sharp_shuffled_n200_p0_acts"].tolist(), d["start_sharp_shuffled_n200_p0_visits"].tolist()))
    assert [by_id[a] for a in acts.tolist()] != visits.tolist()


def test_decode_board_under_another_piece_type_numbering(golden):
    """decode_board (tools.py:74-106) as the reference computes it when the rules module numbers KING 1 .. PAWN 7:
    the oracle's plane map (twin of ccz_config.plane_of_type) reproduces the reference's planes."""
    d = golden["data"]
    pot = golden["meta"]["decode_alt_plane_of_type"]
    try:
        oracle.set_rules(plane_of_type=pot)
        for nm in ("start", "wide80"):
            b = OracleBoard.from_array(d[f"decode_alt_{nm}_sq"], 1, 0)
            red, black = b.decode()
            assert np.array_equal(red, d[f"decode_alt_{nm}_red"]) and np.array_equal(black, d[f"decode_alt_{nm}_black"])
            planes = b.leaf_planes()
            assert np.array_equal(planes[7], d[f"decode_alt_{nm}_red"]) and np.array_equal(planes[15], d[f"decode_alt_{nm}_black"])
    finally:
        oracle.set_rules()
    red, black = OracleBoard().decode()
    assert not np.array_equal(red, d["decode_alt_start_red"])  # the default numbering gives other planes


def test_float16_value_changes_the_search(golden):
    """The float16-value traces differ from what float32 arithmetic gives on the same inputs (the dtype is load-bearing),
    and every stored Q is a float16-representable number."""
    case = [c for c in golden["meta"]["cases"] if c["name"] == "start_sharp_f16value_n300"][0]
    d = golden["data"]
    q = d["start_sharp_f16value_n300_p0_q"]
    assert np.array_equal(q, q.astype(np.float16).astype(np.float32))
    m32 = OracleMCTS(_make_eval(case["ev"]), c_puct=5, n_playout=case["n"])
    m32.get_move_probs(OracleBoard(), 1.0)
    _, v32, q32, _ = m32.root_children()
    assert not np.array_equal(q32, q)


def test_host_twins_of_the_lockstep_entry_points_on_the_cpu():
    """oracle/ccz_ref.c (``ccz_ref_*``: the signatures of include/cczero.h on host memory) is the sequential search behind a lockstep
    call surface: B boards stepped through select / expand+backup / finish_move give what B separate OracleMCTS objects give, the
    sampled move is the Philox twin's, and the config struct it takes is the header's (same layout as the product's ctypes mirror).
    The same object is compared with the HIP library byte for byte in tests/test_gpu_ref_twins.py."""
    import ctypes
    import oracle
    from oracle import OracleBoard, OracleMCTS, RefEngine
    from oracle.evaluators import hash_eval
    from chinesechesszero_amd import _lib
    assert ctypes.sizeof(oracle.RefConfig) == ctypes.sizeof(_lib.Config) == 96
    for (a, _), (b, _) in zip(oracle.RefConfig._fields_, _lib.Config._fields_):
        assert a == b and getattr(oracle.RefConfig, a).offset == getattr(_lib.Config, b).offset
    B, n = 3, 48
    e = RefEngine(B, n_playout=n, seed=11, board_id_base=40)
    boards = [OracleBoard() for _ in range(B)]
    trees = [OracleMCTS(None, c_puct=5, n_playout=0) for _ in range(B)]
    try:
        for move in range(3):
            for s in range(n):
                planes = e.select_leaves().astype(np.float32).reshape(B, 17, 7, 90)
                info = e.leaf_info()
                types = np.arange(1, 8)[None, :, None]
                sq = ((planes[:, 7] * types).sum(1) + (planes[:, 15] * (types + 8)).sum(1)).astype(np.uint8)
                turn = planes[:, 16, 0, 0].astype(np.uint8)
                P, V = np.zeros((B, 2086), np.float32), np.zeros(B, np.float32)
                for b in range(B):
                    P[b], V[b] = (x[0] for x in hash_eval(sq[b:b + 1], turn[b:b + 1], salt=b))
                    leaf, depth = trees[b].select(boards[b])
                    ids = leaf.legal_ids()
                    assert depth == info["depth"][b] and info["ids"][b][:info["k"][b]].tolist() == ids and np.array_equal(leaf.squares(), sq[b])
                    trees[b].expand_backup(leaf, ids, P[b][ids], V[b])
                e.expand_backup(P, V)
            rc = e.root_children()
            for b in range(B):
                acts, visits, q, prior = trees[b].root_children()
                k = len(acts)
                assert rc["k"][b] == k and np.array_equal(rc["acts"][b][:k], acts.astype(np.uint16)) and np.array_equal(rc["visits"][b][:k], visits)
                assert np.array_equal(rc["q"][b][:k].view(np.uint32), q.view(np.uint32)) and rc["root_visits"][b] == trees[b].root_visits()
            moves = e.finish_move()
            for b in range(B):
                acts, visits, _, _ = trees[b].root_children()
                temp = 1.0 if move + 1 <= 30 else 0.5
                want = int(acts[oracle.det_sample(11, 40 + b, move, oracle.det_pi(visits, temp), 0.25, 0.2)[0]])
                assert moves[b] == want
                trees[b].update_with_move(want)
                boards[b].push_id(want)
            st = e.game_status()
            assert st["plies"].tolist() == [move + 1] * B and np.array_equal(e.root_positions(), np.stack([bd.squares() for bd in boards]))
    finally:
        e.close()
