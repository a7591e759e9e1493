"""GPU: lockstep PUCT search of the HIP engine against the oracle and the reference's golden traces."""
import numpy as np
import pytest

from golden_cases import case_start

pytestmark = pytest.mark.gpu


def _engine(B, n, **kw):
    from chinesechesszero_amd.engine import SelfPlayEngine
    # parity tests: a pruned subtree or an adjudicated game is an error, not a counter (CCZ_FLAG_STRICT) -- unless the test sets a ply cap
    # or a node budget on purpose
    kw.setdefault("strict", not ({"max_plies", "max_nodes", "reserve_nodes"} & set(kw)))
    return SelfPlayEngine(B, n_playout=n, **kw)


def test_lockstep_matches_oracle_every_step():
    """8 boards, distinct evaluators, 3 plies x 96 sims: leaf, legal ids, status, planes, then N/Q/P bit-exact."""
    from gpu_harness import Lockstep
    from oracle import OracleBoard
    B = 8
    e = _engine(B, 96)
    ls = Lockstep(e, [OracleBoard() for _ in range(B)], kind="hash_sharp", salts=list(range(100, 100 + B)))
    rs = np.random.RandomState(0)
    for ply in range(3):
        for _ in range(96):
            ls.step(check_leaf=True)
        rc = ls.compare_roots()
        moves = []
        for b in range(B):
            k = rc["k"][b]
            w = rc["visits"][b][:k].astype(np.float64) + 1e-3
            moves.append(int(rc["acts"][b][rs.choice(k, p=w / w.sum())]))
        ls.play(moves)
        assert np.array_equal(e.root_positions(), np.stack([x.squares() for x in ls.boards]))
    ls.compare_roots()
    e.check_healthy()


def test_fused_step_kernel_matches_oracle():
    """The fused expand_backup+select launch (ccz_step) walks exactly the same trees as the two-kernel form."""
    from gpu_harness import Lockstep
    from oracle import OracleBoard
    from golden_cases import STARTS
    B = 6
    e = _engine(B, 80)
    boards = [OracleBoard(), OracleBoard(), OracleBoard.from_array(STARTS["two_rooks"], 1, 112),
              OracleBoard.from_array(STARTS["rook_knight"], 0, 100), OracleBoard.from_array(STARTS["capture_to_bare"], 1, 0),
              OracleBoard()]
    for b in (2, 3, 4):
        e.set_position(b, boards[b].squares(), 1 if boards[b].turn else 0, boards[b].halfmove)
    ls = Lockstep(e, boards, kind="hash_sharp", salts=[5, 6, 7, 8, 9, 10])
    for ply in range(2):
        ls.run_fused(80, check_leaf=True)
        rc = ls.compare_roots()
        over = e.game_status()["over"]
        ls.play([-1 if (over[b] or rc["k"][b] == 0) else int(rc["acts"][b][int(np.argmax(rc["visits"][b][:rc["k"][b]]))])
                 for b in range(B)])
    ls.compare_roots()
    st = e.stats()
    assert st["sims"] >= 3 * 160 and st["terminal_leaves"] > 0
    e.check_healthy()


def test_uniform_priors_exact_ties_first_max_order():
    from gpu_harness import Lockstep
    from oracle import OracleBoard
    e = _engine(2, 150)
    ls = Lockstep(e, [OracleBoard(), OracleBoard()], kind="uniform")
    for _ in range(150):
        ls.step(check_leaf=True)
    rc = ls.compare_roots()
    assert rc["root_visits"][0] == 150 and rc["visits"][0][:44].sum() == 149  # first playout expands the root
    e.check_healthy()


@pytest.mark.parametrize("idx", range(20))
def test_golden_traces_from_reference_mcts(golden, idx, rules_of_case):
    """Visit counts / Q / priors equal the numbers the reference's own mcts.py produced (bit-exact),
    pi within 1e-12, with the golden moves forced (tree reuse across plies)."""
    from gpu_harness import Lockstep
    from oracle import OracleBoard
    case = golden["meta"]["cases"][idx]
    d = golden["data"]
    name = case["name"]
    sqs, turn, half = case_start(case)
    rank, trank = rules_of_case(case, both=True)  # shuffled / type-major `legal_moves` orders through ccz_config.move_rank_host, type_rank
    f16 = case.get("value_dtype") == "float16"    # cases 15, 16: the reference's CUDA-path Q dtype (CCZ_FLAG_VALUE_F16)
    e = _engine(1, case["n"], move_rank=rank, type_rank=trank, value_f16=f16)
    if case["start"] != "start":
        e.set_position(0, sqs, turn, half)
        ob = OracleBoard.from_array(sqs, turn, half)
    else:
        ob = OracleBoard()
    salt = {"hash": 0, "hash_sharp": 7, "uniform": 0}[case["ev"]]
    ls = Lockstep(e, [ob], kind=case["ev"], salts=[salt], value_f16=f16)
    for ply in range(case["plies_done"]):
        if idx % 2 == 0:   # even cases through the fused launch sequence, odd ones through select + expand_backup
            ls.run_fused(case["n"], check_leaf=False)
        else:
            for _ in range(case["n"]):
                ls.step(check_leaf=False)
        rc = e.root_children()
        k = int(rc["k"][0])
        assert np.array_equal(rc["acts"][0][:k], d[f"{name}_p{ply}_acts"].astype(np.uint16))
        assert np.array_equal(rc["visits"][0][:k], d[f"{name}_p{ply}_visits"])
        assert np.array_equal(rc["q"][0][:k].view(np.uint32), d[f"{name}_p{ply}_q"].view(np.uint32))
        assert np.array_equal(rc["prior"][0][:k].view(np.uint32), d[f"{name}_p{ply}_prior"].view(np.uint32))
        assert rc["root_visits"][0] == int(d[f"{name}_p{ply}_rootvisits"])
        pi = e.root_pi(temps=case["temps"][ply])[0][:k]
        assert np.allclose(pi, d[f"{name}_p{ply}_pi"], rtol=0, atol=1e-12)
        move = int(d[f"{name}_p{ply}_move"])
        if case["selfplay"]:
            ls.play([move])
        else:  # match play discards the tree (mcts.py:228-229)
            e.finish_move(forced_moves=np.array([move], np.int32), keep_tree=False)
            ls.mcts[0].update_with_move(-1)
            ls.boards[0].push_id(move)
    assert np.array_equal(e.root_positions()[0], d[f"{name}_final_sq"])
    e.check_healthy()


def test_device_sampler_matches_cpu_twin():
    """pi (deterministic log/exp) and the Dirichlet-mixed Philox choice: bit-exact vs oracle/xq_sample.c."""
    import oracle
    from gpu_harness import Lockstep
    from oracle import OracleBoard
    B = 16
    e = _engine(B, 40, seed=1234, board_id_base=5000)
    ls = Lockstep(e, [OracleBoard() for _ in range(B)], kind="hash_sharp", salts=list(range(B)))
    for ply in range(4):
        for _ in range(40):
            ls.step(check_leaf=False)
        rc = ls.compare_roots()
        temp = 1.0 if ply < 2 else 0.5
        pi = e.root_pi(temps=temp)
        moves = e.finish_move(temps=np.full(B, temp)).cpu().numpy()
        for b in range(B):
            k = int(rc["k"][b])
            want_pi = oracle.det_pi(rc["visits"][b][:k], temp)
            assert np.array_equal(pi[b][:k].view(np.uint64), want_pi.view(np.uint64)), (ply, b)
            idx, _ = oracle.det_sample(1234, 5000 + b, ply, want_pi, 0.25, 0.2)
            assert moves[b] == rc["acts"][b][idx], (ply, b, moves[b], rc["acts"][b][idx])
            ls.mcts[b].update_with_move(int(moves[b]))
            ls.boards[b].push_id(int(moves[b]))
    ls.compare_roots()
    e.check_healthy()


def test_device_sampler_distribution():
    """Sampled moves follow (1-eps)*pi + eps*Dirichlet in aggregate: chi-square-like sanity on 2048 boards."""
    import torch
    B = 2048
    e = _engine(B, 8, seed=99)
    P = torch.full((B, 2086), 1.0 / 2086, dtype=torch.float32, device=e.device)
    V = torch.zeros(B, dtype=torch.float32, device=e.device)
    for _ in range(60):
        e.select_leaves()
        e.expand_backup(P, V)
    rc = e.root_children()
    assert np.all(rc["k"] == 44) and np.all(rc["visits"] == rc["visits"][0])
    pi = e.root_pi(temps=1.0)[0][:44]
    moves = e.finish_move().cpu().numpy()
    counts = np.array([(moves == a).sum() for a in rc["acts"][0][:44]], np.float64)
    expect = B * (0.75 * pi + 0.25 / 44)
    assert counts.sum() == B
    assert np.all(np.abs(counts - expect) < 6 * np.sqrt(expect) + 6)
    assert len(np.unique(moves)) > 20
    e.check_healthy()


def _legal_count(sq, turn):
    from oracle import OracleBoard
    return len(OracleBoard.from_array(sq, turn, 0).legal_ids())


def test_wide_nodes_more_than_64_children():
    """Positions with > 64 legal moves: second 64-lane pass of select / expand / sampling, big re-root blocks."""
    from gpu_harness import Lockstep
    from golden_cases import sq as S
    from oracle import OracleBoard
    import oracle
    b = np.zeros(90, np.uint8)
    for name, pc in {"e1": 7, "a2": 3, "i7": 3, "b4": 2, "h5": 2, "c3": 4, "g6": 4, "a6": 1, "c7": 1, "e6": 1, "g7": 1, "i6": 1,
                     "d0": 6, "f0": 6, "c0": 5, "g0": 5, "d9": 15, "e8": 14, "a9": 11}.items():
        b[S(name)] = pc
    k0 = _legal_count(b, 1)
    assert k0 > 64, k0
    B = 3
    e = _engine(B, 160, seed=8)
    boards = []
    for i in range(B):
        e.set_position(i, b, 1, 0)
        boards.append(OracleBoard.from_array(b, 1, 0))
    ls = Lockstep(e, boards, kind="hash_sharp", salts=[31, 32, 33])
    for ply in range(2):
        ls.run_fused(160, check_leaf=True)
        rc = ls.compare_roots()
        if ply == 0:
            assert rc["k"][0] == k0
            # device sampler over k > 64 children agrees with its CPU twin
            pi = e.root_pi(temps=1.0)
            moves = e.finish_move(temps=np.ones(B)).cpu().numpy()
            for i in range(B):
                k = int(rc["k"][i])
                want = oracle.det_pi(rc["visits"][i][:k], 1.0)
                assert np.array_equal(pi[i][:k].view(np.uint64), want.view(np.uint64))
                idx, _ = oracle.det_sample(8, i, 0, want, 0.25, 0.2)
                assert moves[i] == rc["acts"][i][idx]
                ls.mcts[i].update_with_move(int(moves[i]))
                ls.boards[i].push_id(int(moves[i]))
    ls.compare_roots()
    e.check_healthy()


def test_deep_paths_beyond_one_wave():
    """Selection paths deeper than 64 plies: chunked path replay, backup lanes wrapping, repetition / sixty-move leaves."""
    import torch
    from gpu_harness import planes_to_squares
    from golden_cases import sq as S
    from oracle import OracleBoard, OracleMCTS
    from oracle.evaluators import position_hash
    b = np.zeros(90, np.uint8)
    # kings + five pawns each: pawn pushes are irreversible, so long lines do not die of repetition early
    for name, pc in {"d0": 7, "a3": 1, "c3": 1, "e3": 1, "g3": 1, "i3": 1, "f9": 15, "a6": 9, "c6": 9, "e6": 9, "g6": 9, "i6": 9}.items():
        b[S(name)] = pc
    B, n = 2, 1000
    e = _engine(B, n, seed=2)
    boards, mcts = [], []
    for i in range(B):
        e.set_position(i, b, 1 - i, 0)
        boards.append(OracleBoard.from_array(b, 1 - i, 0))
        mcts.append(OracleMCTS(None, c_puct=5, n_playout=0))

    def evaluate(sq, turn, salt):
        """prior mass 0.97 on one hash-chosen legal move: PUCT digs one long principal line"""
        ob = OracleBoard.from_array(sq, int(turn), 0)
        ids = ob.legal_ids()
        P = np.full(2086, 0.0, np.float32)
        if ids:
            P[ids] = np.float32(0.03 / len(ids))
            P[ids[int(position_hash(sq[None, :], np.array([turn]), salt)[0] % np.uint64(len(ids)))]] = np.float32(0.97)
        return P, np.float32(0.0)

    e.select_leaves()
    for it in range(n):
        planes = e.leaf_input.float().cpu().numpy()
        info = e.leaf_info()
        sq, turn = planes_to_squares(planes)
        P = np.zeros((B, 2086), np.float32)
        V = np.zeros(B, np.float32)
        for i in range(B):
            leaf, depth = mcts[i].select(boards[i])
            assert depth == info["depth"][i] and np.array_equal(leaf.squares(), sq[i]), (it, i, depth, info["depth"][i])
            end, tie = leaf.is_game_over(), leaf.is_tie()
            assert info["status"][i] == (0 if (not end and not tie) else (1 if (end and tie) else 2)), (it, i)
            P[i], V[i] = evaluate(sq[i], turn[i], 5 + i)
            ids = leaf.legal_ids()
            mcts[i].expand_backup(leaf, ids, P[i][ids], V[i])
        tp, tv = torch.from_numpy(P).to(e.device), torch.from_numpy(V).to(e.device)
        if it + 1 < n:
            e.step(tp, tv)
        else:
            e.expand_backup(tp, tv)
    st = e.stats()
    assert st["depth_peak"] >= 64, st["depth_peak"]
    assert st["terminal_leaves"] > 0
    rc = e.root_children()
    for i in range(B):
        acts, visits, q, prior = mcts[i].root_children()
        k = len(acts)
        assert np.array_equal(rc["visits"][i][:k], visits)
        assert np.array_equal(rc["q"][i][:k].view(np.uint32), q.view(np.uint32))
    e.check_healthy()


def test_long_quiet_history_chain_and_sixty_move_rule():
    """> 64 plies without a capture: the history-chain tail beyond the prefetched 64 keys, repetition counting over
    it, and the sixty-move rule ending the game exactly when the oracle says so."""
    from gpu_harness import Lockstep
    from oracle import OracleBoard
    rs = np.random.RandomState(21)
    B = 2
    e = _engine(B, 60, seed=4)
    boards = [OracleBoard() for _ in range(B)]
    # a quiet random walk: no captures, never ending the game, 100 plies
    for ply in range(100):
        forced = []
        for b in range(B):
            ob = boards[b]
            cands = []
            for m in ob.legal_ids():
                c = ob.copy()
                was = c.halfmove
                c.push_id(m)
                if c.halfmove == was + 1 and not c.is_game_over():
                    cands.append(m)
            m = cands[rs.randint(len(cands))]
            forced.append(m)
            ob.push_id(m)
        e.finish_move(forced_moves=np.array(forced, np.int32), keep_tree=False)
    assert all(ob.halfmove == 100 for ob in boards)
    assert np.array_equal(e.root_positions(), np.stack([ob.squares() for ob in boards]))
    ls = Lockstep(e, boards, kind="hash_sharp", salts=[61, 62])
    ls.run_fused(60, check_leaf=True)      # leaf status (repetition over the long chain) checked at every step
    ls.compare_roots()
    # keep playing quiet moves: the engine must flag the game over exactly at the oracle's sixty-move point
    over_at = [None] * B
    for ply in range(30):
        forced = np.full(B, -1, np.int32)
        for b in range(B):
            ob = boards[b]
            if over_at[b] is not None:
                continue
            cands = [m for m in ob.legal_ids() if ob.squares()[oracle_to(m)] == 0]
            m = cands[rs.randint(len(cands))]
            forced[b] = m
            ob.push_id(m)
            ls.mcts[b].update_with_move(-1)
            if ob.is_game_over():
                over_at[b] = ob.halfmove
        e.finish_move(forced_moves=forced, keep_tree=False)
        st = e.game_status()
        for b in range(B):
            assert bool(st["over"][b]) == (over_at[b] is not None), (ply, b, st["over"][b], over_at[b])
        if all(x is not None for x in over_at):
            break
    assert all(x is not None and x <= 120 for x in over_at)
    assert list(e.game_status()["winner"]) == [-1 if boards[b].outcome().winner is None else int(boards[b].outcome().winner) for b in range(B)]
    e.check_healthy()


def oracle_to(mid):
    import oracle
    return oracle.lib().xq_move_to(int(mid))
