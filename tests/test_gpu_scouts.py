"""GPU: one game at a time with scout slots (include/cczero.h ccz_scout, selfplay.ScoutedSearch; round 6).

The reference's first-maximum rule (mcts.py:47-48,59-61) makes the children of a node first-visited in legal_moves order, so the
leaves the search will ask for next are known -- the pending leaf's next siblings -- and can be evaluated in the same evaluator call.
What must hold: the SAME tree, bit for bit, with fewer evaluator calls."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(blocks=2):
    from chinesechesszero_amd.net import PolicyValueNet
    torch.manual_seed(11)
    pvn = PolicyValueNet(device="cuda:0", num_channels=256, resblocks_num=blocks)   # 256 wide: the hand-written evaluator, whose result
    for m in pvn.policy_value_net.modules():                                         # for a row does not depend on the batch it sits in
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    pvn.refresh_inference_copy()
    return pvn


def _search(pvn, scouts, n, plies, use_graph, seed=3, on_playout=None):
    from chinesechesszero_amd.game import Board
    from chinesechesszero_amd.mcts import MCTS_AI
    np.random.seed(seed)
    ai = MCTS_AI(pvn.policy_value_fn, c_puct=5, n_playout=n, is_selfplay=True)
    ai.mcts = type(ai.mcts)(pvn.policy_value_fn, 5, n, scouts=scouts)
    ai.mcts.use_graph = use_graph
    board = Board()
    roots, moves = [], []
    for ply in range(plies):
        acts, probs = ai.mcts.get_move_probs(board, temp=1.0, on_playout=on_playout)
        rc = ai.mcts.root_children()
        roots.append({k: np.array(v).copy() for k, v in rc.items()})
        mv = int(np.random.choice(acts, p=probs))
        ai.mcts.update_with_move(mv)
        board.push(mv)
        moves.append(mv)
    return roots, moves, ai.mcts


@pytest.mark.parametrize("use_graph", [False, True])
def test_a_scouted_search_builds_the_same_tree_with_fewer_evaluator_calls(use_graph):
    pvn = _net()
    n, plies = 96, 5
    want, moves0, plain = _search(pvn, 0, n, plies, use_graph)
    calls = {}
    for scouts in (1, 7, 20):
        got, moves, m = _search(pvn, scouts, n, plies, use_graph)
        assert moves == moves0
        for ply, (a, b) in enumerate(zip(want, got)):
            for key in ("k", "acts", "visits", "q", "prior", "root_visits"):
                assert np.array_equal(a[key], b[key]), (scouts, ply, key)
        s = m._scouted
        assert s.simulations == n * plies and 0 < s.evaluator_calls
        calls[scouts] = s.evaluator_calls
        m._engine.check_healthy()
        st = m._engine.stats()
        assert st["sims"] == n * plies and st["error_flags"] == 0      # the scout slots are not simulated
    assert plain._scouted is None
    # fewer evaluator calls the more scouts (this small net searches narrow and deep -- the worst case: 0.65 / 0.49 / 0.46 calls per
    # simulation; the full 40 x 256 net: 0.24 with 7 scouts, profiles/r06_single_board.json)
    assert calls[20] <= calls[7] < calls[1] < n * plies and calls[7] < 0.55 * n * plies, calls


@pytest.mark.parametrize("use_graph", [False, True])
def test_the_device_side_loop_of_hit_simulations_is_the_host_loop(use_graph, monkeypatch):
    """ccz_scouted_run repeats step + scout + probe + plan on the device while board 0's leaf is in the table; the host loop
    (CCZ_SCOUT_DEVICE_LOOP=0) launches the same phases one simulation at a time. Same trees, same moves, same evaluator calls -- and
    on_playout (mcts.py:154-160) is called at the same playouts with the same counts, whether the budget cuts a run short or not."""
    pvn = _net()
    n, plies = 230, 4           # interval = 2: every run is cut short by the progress report; without a callback: only by misses and the move's end
    out = {}
    for loop in ("0", "1"):
        monkeypatch.setenv("CCZ_SCOUT_DEVICE_LOOP", loop)
        for cb in (False, True):
            seen = []
            roots, moves, m = _search(pvn, 10, n, plies, use_graph, on_playout=(seen.append if cb else None))
            assert m._scouted.device_loop == (loop == "1")
            st = m._engine.stats()
            assert st["sims"] == n * plies and st["error_flags"] == 0
            m._engine.check_healthy()
            out[loop, cb] = (roots, moves, m._scouted.evaluator_calls, m._scouted.simulations, seen)
    want = out["0", False]
    assert want[3] == n * plies and 0 < want[2] < 0.6 * n * plies
    assert out["0", True][4] == [2] * (n // 2 * plies)
    for key, got in out.items():
        assert got[1] == want[1] and got[2] == want[2] and got[3] == want[3], key
        assert got[4] == (out["0", True][4] if key[1] else []), key
        for ply, (a, b) in enumerate(zip(want[0], got[0])):
            for k in ("k", "acts", "visits", "q", "prior", "root_visits"):
                assert np.array_equal(a[k], b[k]), (key, ply, k)


def test_scouted_run_counts_and_stops(monkeypatch):
    """ccz_scouted_run by itself: it does at most `budget` simulations, never more than the move has left, reports a miss with the
    plan in place, and the move's last simulation is backed up without a next selection. Stub evaluator rows (uniform priors,
    value 0) handed to the planned gather, so every count is known."""
    from chinesechesszero_amd.engine import SelfPlayEngine
    B = 8
    e = SelfPlayEngine(B, n_playout=64, seed=1, eval_cache_log2=16, strict=True)
    e.set_scouts(B - 1)
    logits = torch.zeros((B, 2086), dtype=torch.float16, device="cuda:0")
    value = torch.zeros((B,), dtype=torch.float32, device="cuda:0")
    e.select_leaves()
    e.scout_and_plan()
    assert e.plan_state_of_board0() == 0          # the root: not in the (empty) table
    sims, calls, left = 0, 0, 40
    need = True
    while left > 0:
        if need:
            e.gather_priors_planned(logits, value)
            calls += 1
        e.set_run(5, left)
        e.scouted_run_launch()
        done, need = e.run_outcome()
        assert 1 <= done <= min(5, left)
        if done < min(5, left):
            assert need                           # cut short: only a miss does that
        left -= done
        sims += done
    st = e.stats()
    assert sims == 40 and st["sims"] == 40 and st["error_flags"] == 0
    # uniform priors, value 0: the root's children are visited in order, 7 scouts ahead -> one evaluator call answers 8 simulations
    # (the root's own evaluation + ceil(39 / 8) calls for its first 39 children; one more where two of the 40 positions fall on the same
    # table slot -- the loser of a slot is evaluated but not stored)
    assert 1 + -(-39 // 8) <= calls <= 2 + -(-39 // 8), calls
    rc = e.root_children()
    assert int(rc["root_visits"][0]) == 40 and rc["visits"][0][:39].tolist() == [1] * 39 and int(rc["visits"][0][39]) == 0
    with pytest.raises(ValueError):
        e.set_run(0, 1)
    e.check_healthy()


def test_scouted_run_with_two_searched_boards_is_the_separate_launches():
    """The device-side loop is written for any number of searched boards (it leaves the loop when ANY of them misses). Two boards on
    different start positions, six scout slots, a deterministic evaluator whose result for a row depends on that row only (a fixed random
    linear map of the input planes): ccz_scouted_run with ragged budgets against step + scout_and_plan one simulation at a time -- same
    trees, same number of evaluator calls."""
    from golden_cases import STARTS
    from chinesechesszero_amd.engine import SelfPlayEngine
    B, active, n = 8, 2, 150
    g = torch.Generator(device="cpu").manual_seed(5)
    wp = (torch.randn((17 * 7 * 90, 2086), generator=g) * 0.05).to("cuda:0", torch.float16)
    wv = (torch.randn((17 * 7 * 90,), generator=g) * 0.02).to("cuda:0", torch.float16)

    def evaluator(leaf):
        x = leaf.reshape(leaf.shape[0], -1)
        return (x @ wp).contiguous(), torch.tanh((x.float() @ wv.float())).contiguous()

    def engine():
        e = SelfPlayEngine(B, n_playout=n, seed=4, eval_cache_log2=14, strict=True)
        e.set_scouts(B - active)
        e.set_position(1, STARTS["rook_knight"].copy(), 1, 0)
        e.select_leaves()
        e.scout_and_plan()
        return e

    a, calls_a = engine(), 0
    for sim in range(n):
        if (a.plan_states() == 0).any():
            a.gather_priors_planned(*evaluator(a.leaf_input))
            calls_a += 1
        if sim + 1 == n:
            a.expand_backup_compact(None)
        else:
            a.step_compact(None)
            a.scout_and_plan()
    b, calls_b, left, k = engine(), 0, n, 0
    need = bool((b.plan_states() == 0).any())
    while left > 0:
        if need:
            b.gather_priors_planned(*evaluator(b.leaf_input))
            calls_b += 1
        b.set_run((3, 1, 1000, 7)[k % 4], left)
        k += 1
        b.scouted_run_launch()
        done, need = b.run_outcome()
        left -= done
    assert calls_a == calls_b and 0 < calls_a < n
    ra, rb = a.root_children(), b.root_children()
    for key in ("k", "acts", "visits", "q", "prior", "root_visits"):
        assert np.array_equal(ra[key][:active], rb[key][:active]), key
    assert int(ra["root_visits"][0]) == n and int(ra["root_visits"][1]) == n
    sa, sb = a.stats(), b.stats()
    for key in ("sims", "expansions", "sum_children", "sum_depth", "nodes_peak", "error_flags"):
        assert sa[key] == sb[key], key
    assert sa["sims"] == active * n and sa["error_flags"] == 0
    a.check_healthy()
    b.check_healthy()


@pytest.mark.parametrize("device_loop", [False, True])
def test_what_the_table_returns_is_what_the_evaluator_returns_under_scouts(device_loop):
    """CCZ_FLAG_CACHE_VERIFY on a scouted engine: one table hit in 128 is planned as an evaluator row all the same (for the device-side
    loop that is a miss: it leaves the loop, the scouts are handed their leaves, the evaluator runs) and the fresh priors / value are
    compared with the cached ones bit for bit -- the assumption the scouts rest on (the evaluator's result for a row does not depend on
    the batch it sits in, nor on which call computed it), checked on the real hand-written evaluator."""
    from chinesechesszero_amd.engine import SelfPlayEngine
    from chinesechesszero_amd.selfplay import ScoutedSearch
    pvn = _net()
    n, moves = 400, 6
    e = SelfPlayEngine(11, n_playout=n, seed=9, eval_cache_log2=16, cache_verify=True, strict=True)
    e.set_scouts(10)
    s = ScoutedSearch(e, pvn.evaluate_leaves_logits, use_graph=True, device_loop=device_loop)
    forced = np.full(11, -1, np.int32)
    for _ in range(moves):
        s.begin_move()
        left = n
        while left > 0:
            if device_loop:
                left -= s.run(left, left)
            else:
                s.simulate(last=left == 1)
                left -= 1
        rc = e.root_children()
        forced[0] = int(rc["acts"][0][int(np.argmax(rc["visits"][0][:int(rc["k"][0])]))])
        e.finish_move(forced_moves=forced, keep_tree=True)
    st = e.stats()
    assert s.simulations == n * moves and st["sims"] == n * moves and st["error_flags"] == 0
    assert st["cache_verified"] >= 5 and st["cache_verify_mismatches"] == 0, st
    e.check_healthy()


def test_scout_slots_hold_the_next_unvisited_siblings_of_the_pending_leaf():
    """ccz_scout by itself. With uniform priors and value 0 every PUCT comparison is a tie, so the search visits the root's children
    in order: while the pending leaf of board 0 is child i of the root, slot j must hold child i + j -- legal moves, status and
    evaluator input as the ORACLE computes them for that position -- or nothing when the root has no such child."""
    from oracle import OracleBoard
    from chinesechesszero_amd.engine import SelfPlayEngine
    from chinesechesszero_amd.net import uniform_evaluator
    B = 6
    e = SelfPlayEngine(B, n_playout=64, seed=1, eval_cache_log2=12, strict=True)
    e.set_scouts(B - 1)
    leaf = e.select_leaves()
    prob, value = uniform_evaluator(leaf)
    e.scout()
    assert (e.leaf_info()["status"][1:] == 3).all()          # the pending leaf is the root itself: no siblings
    root_ids = OracleBoard().legal_ids()
    checked = 0
    for sim in range(44):
        e.step(prob, value)
        e.scout()
        info = e.leaf_info()
        assert info["depth"][0] == 1 and info["status"][0] == 0
        planes = e.leaf_input.float().cpu().numpy().reshape(B, -1)
        i = sim                                                # children 0 .. sim - 1 are expanded, child `sim` is pending
        lead = OracleBoard()
        lead.push_id(root_ids[i])
        assert info["ids"][0][:info["k"][0]].tolist() == lead.legal_ids()
        for j in range(1, B):
            if i + j >= len(root_ids):
                assert info["status"][j] == 3
                continue
            sib = OracleBoard()
            sib.push_id(root_ids[i + j])
            assert info["status"][j] == 0 and info["k"][j] == len(sib.legal_ids())
            assert info["ids"][j][:info["k"][j]].tolist() == sib.legal_ids()
            assert np.array_equal(planes[j], np.asarray(sib.leaf_planes(), np.float32).reshape(-1))
            checked += 1
    assert checked == sum(min(B - 1, 43 - s) for s in range(44))
    assert e.stats()["error_flags"] == 0
    e.check_healthy()


def test_scout_slots_of_two_searched_boards():
    """More than one searched board: slot active + j * active + r scouts for board r, j + 1 children ahead (the host loop of the package
    searches one board; the kernels are written for any number). Two boards on different start positions, uniform evaluator."""
    from golden_cases import STARTS
    from oracle import OracleBoard
    from chinesechesszero_amd.engine import SelfPlayEngine
    from chinesechesszero_amd.net import uniform_evaluator
    B, active = 8, 2
    e = SelfPlayEngine(B, n_playout=64, seed=2, eval_cache_log2=12, strict=True)
    e.set_scouts(B - active)
    starts = [None, (STARTS["rook_knight"].copy(), 1, 0)]
    e.set_position(1, *starts[1])
    roots = [OracleBoard(), OracleBoard.from_array(*starts[1])]
    root_ids = [r.legal_ids() for r in roots]
    leaf = e.select_leaves()
    prob, value = uniform_evaluator(leaf)
    checked = 0
    for sim in range(12):
        e.step(prob, value)
        e.scout_and_plan()                       # the fused launch (8 slots <= 16): scout + probe + plan
        assert e.plan_state_of_board0() in (0, 1)
        info = e.leaf_info()
        planes = e.leaf_input.float().cpu().numpy().reshape(B, -1)
        for q in range(B - active):
            r, ahead, slot = q % active, 1 + q // active, active + q
            i = sim                              # child `sim` of board r's root is pending (children are first visited in order)
            if i + ahead >= len(root_ids[r]):
                assert info["status"][slot] == 3
                continue
            sib = OracleBoard(roots[r].b)
            sib.push_id(root_ids[r][i + ahead])
            assert info["status"][slot] == 0 and info["ids"][slot][:info["k"][slot]].tolist() == sib.legal_ids(), (sim, q)
            assert np.array_equal(planes[slot], np.asarray(sib.leaf_planes(), np.float32).reshape(-1))
            checked += 1
    assert checked == 12 * (B - active)
    e.check_healthy()
