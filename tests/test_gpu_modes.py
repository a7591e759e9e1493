"""GPU: reference-exact NumPy sampling mode of the batched driver, and the loud-failure paths of the engine."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class HashEvaluator:
    """Device-side wrapper of the integer hash evaluator (round trip through the host; small B only)."""

    batched = True

    def __init__(self, salts, scale=40.0):
        self.salts = salts
        self.scale = scale

    def __call__(self, leaf):
        from gpu_harness import planes_to_squares
        from oracle.evaluators import hash_eval
        sq, turn = planes_to_squares(leaf.float().cpu().numpy())
        P = np.zeros((len(sq), 2086), np.float32)
        V = np.zeros(len(sq), np.float32)
        for b in range(len(sq)):
            p, v = hash_eval(sq[b:b + 1], turn[b:b + 1], salt=self.salts[b], scale=self.scale)
            P[b], V[b] = p[0], v[0]
        return torch.from_numpy(P).to(leaf.device), torch.from_numpy(V).to(leaf.device)


def test_numpy_sampling_mode_reproduces_sequential_reference_games():
    """BatchedSelfPlay(sampling='numpy'): every board plays exactly the game a sequential reference-style loop
    (oracle MCTS + game.py's temperature schedule + mcts.py:216-224 sampling on its own RandomState) plays."""
    from oracle import OracleBoard, OracleMCTS
    from oracle.evaluators import hash_eval
    from chinesechesszero_amd.selfplay import BatchedSelfPlay
    B, n, plies, seed = 5, 48, 5, 77
    salts = [21, 22, 23, 24, 25]
    sp = BatchedSelfPlay(HashEvaluator(salts), B, n_playout=n, seed=seed, sampling="numpy")
    got = [[] for _ in range(B)]
    for _ in range(plies):
        mv = sp.run_move().cpu().numpy()
        for b in range(B):
            got[b].append(int(mv[b]))
    sp.engine.check_healthy()
    for b in range(B):
        def ev(board, ids, _s=salts[b]):
            p, v = hash_eval(board.squares()[None, :], np.array([1 if board.turn else 0]), salt=_s, scale=40.0)
            return p[0][ids], v[0]
        board, mcts = OracleBoard(), OracleMCTS(ev, c_puct=5, n_playout=n)
        rs = np.random.RandomState((seed + b) % (2**32))
        want = []
        for ply in range(plies):
            temp = 1.0 if ply + 1 <= 30 else 0.5
            acts, visits, _ = mcts.get_move_probs(board, temp)
            x = 1.0 / temp * np.log(visits.astype(np.int64) + 1e-10)
            probs = np.exp(x - np.max(x))
            probs /= probs.sum()
            move = int(rs.choice(acts, p=0.75 * probs + 0.25 * rs.dirichlet(0.2 * np.ones(len(probs)))))
            mcts.update_with_move(move)
            board.push_id(move)
            want.append(move)
        assert got[b] == want, (b, got[b], want)


def test_node_pool_exhaustion_is_loud_and_safe():
    from chinesechesszero_amd._lib import CczError
    from chinesechesszero_amd.engine import SelfPlayEngine
    from chinesechesszero_amd.net import uniform_evaluator
    e = SelfPlayEngine(4, n_playout=64, max_nodes=200)  # room for ~4 expansions only
    for _ in range(64):
        leaf = e.select_leaves()
        e.expand_backup(*uniform_evaluator(leaf))
    st = e.stats()
    assert st["error_flags"] & 1 and st["nodes_peak"] <= 200 and st["sims"] == 4 * 64
    with pytest.raises(CczError, match="node pool"):
        e.check_healthy()
    # the engine stays usable: a move can still be played from the visits it has
    moves = e.finish_move().cpu().numpy()
    assert np.all(moves >= 0)


def test_nan_priors_and_bad_arguments_are_rejected():
    from chinesechesszero_amd._lib import CczError
    from chinesechesszero_amd.engine import SelfPlayEngine
    e = SelfPlayEngine(2, n_playout=8)
    leaf = e.select_leaves()
    P = torch.full((2, 2086), float("nan"), device=e.device)
    V = torch.zeros(2, device=e.device)
    e.expand_backup(P, V)
    for _ in range(50):  # unvisited children score +inf whatever their prior; once all 44 are visited every score is NaN
        e.select_leaves()
        e.expand_backup(P, V)
    assert e.stats()["error_flags"] & 32
    with pytest.raises(TypeError):
        e.expand_backup(P.double(), V)
    with pytest.raises(ValueError):
        e.expand_backup(P[:1], V)
    with pytest.raises(ValueError):
        e.expand_backup(P.cpu(), V.cpu())
    with pytest.raises(CczError):
        e.set_position(0, np.zeros(90, np.uint8), 1, 0)  # no kings
    with pytest.raises(CczError):
        e.set_position(5, np.zeros(90, np.uint8), 1, 0)  # board index out of range
    with pytest.raises(CczError):
        SelfPlayEngine(2, max_depth=100000)


def test_concurrent_trainer_config5_smoke():
    """bench.py --train-every (BASELINE config 5 on one GPU): self-play keeps running next to trainer updates."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--boards", "64", "--playout", "8", "--steps", "24", "--warmup", "2",
                          "--blocks", "2", "--channels", "32", "--train-every", "6", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["trainer_updates"] == 4 and j["value"] > 0 and j["n_gpus"] == 1
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in j
    assert set(j["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}


def test_hipgraph_replay_equals_eager_launches():
    """Replaying (evaluator, k_step) as one hipGraph walks the same trees as eager launches; small-B timing printed."""
    import time
    from chinesechesszero_amd.net import PolicyValueNet
    from chinesechesszero_amd.selfplay import BatchedSelfPlay
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    pvn = PolicyValueNet(device=dev, num_channels=32, resblocks_num=4)
    res = {}
    for use_graph in (False, True):
        sp = BatchedSelfPlay(pvn.evaluate_leaves, 8, n_playout=40, seed=3, use_graph=use_graph)
        sp.run_move()                       # first move also pays capture / warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        moves = sp.run_move().cpu().numpy()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rc = sp.engine.root_children()
        sp.engine.check_healthy()
        res[use_graph] = (moves, rc["visits"].copy(), rc["acts"].copy(), sp.engine.root_positions(), dt)
    assert np.array_equal(res[False][0], res[True][0])
    assert np.array_equal(res[False][1], res[True][1]) and np.array_equal(res[False][2], res[True][2])
    assert np.array_equal(res[False][3], res[True][3])
    print(f"40 sims x 8 boards: eager {res[False][4] * 1e3:.1f} ms, hipGraph {res[True][4] * 1e3:.1f} ms")


def test_logits_boundary_gathers_the_same_priors():
    """ccz_gather_priors + compact step == exp(log_softmax(logits))[legal ids] fed through the dense boundary."""
    from chinesechesszero_amd.engine import SelfPlayEngine
    dev = torch.device("cuda", 0)
    B = 64
    g = torch.Generator(device=dev).manual_seed(5)
    for dtype in (torch.float32, torch.float16):
        ea, eb = SelfPlayEngine(B, n_playout=16, seed=1), SelfPlayEngine(B, n_playout=16, seed=1)
        la, lb = ea.select_leaves(), eb.select_leaves()
        for it in range(12):
            logits = (torch.randn((B, 2086), device=dev, generator=g) * 3).to(dtype).contiguous()
            value = torch.tanh(torch.randn(B, device=dev, generator=g)).contiguous()
            prob = torch.exp(torch.log_softmax(logits.float(), dim=1)).contiguous()
            la = ea.step_logits(logits, value)
            lb = eb.step(prob, value)
            assert torch.equal(la, lb), it          # same leaves selected (priors agree far below PUCT gaps here)
        ra, rb = ea.root_children(), eb.root_children()
        assert np.array_equal(ra["k"], rb["k"]) and np.array_equal(ra["acts"], rb["acts"])
        assert np.array_equal(ra["visits"], rb["visits"])
        assert np.allclose(ra["prior"], rb["prior"], rtol=2e-6, atol=1e-12)
        assert np.allclose(ra["q"], rb["q"], rtol=0, atol=1e-6)
        ea.check_healthy()
        eb.check_healthy()


def test_reroot_budget_prunes_deepest_levels_instead_of_failing():
    """A kept subtree larger than the pool budget loses its deepest children at re-root time (they become leaves
    again); the top of the tree, the visit counts of the new root and the health flags are untouched."""
    from chinesechesszero_amd.engine import SelfPlayEngine
    from test_gpu_soak import LinearEvaluator
    B, n = 6, 300
    ev = LinearEvaluator(torch.device("cuda", 0), seed=2, sharp=14.0)
    e = SelfPlayEngine(B, n_playout=n, max_nodes=40000, reserve_nodes=37000, seed=3)   # kept subtree budget: 3000 nodes
    pruned_before = 0
    for move in range(6):
        leaf = e.select_leaves()
        for i in range(n):
            p, v = ev(leaf)
            if i + 1 < n:
                leaf = e.step(p, v)
            else:
                e.expand_backup(p, v)
        rc = e.root_children()
        best = [int(np.argmax(rc["visits"][b][:rc["k"][b]])) for b in range(B)]      # keep as much of the tree as possible
        forced = np.array([rc["acts"][b][best[b]] for b in range(B)], np.int32)
        kept_visits = np.array([rc["visits"][b][best[b]] for b in range(B)])
        e.finish_move(forced_moves=forced)
        after = e.root_children()
        assert np.array_equal(after["root_visits"], kept_visits)                       # the new root keeps its statistics
        st = e.stats()
        assert st["error_flags"] == 0, st
        assert st["nodes_peak"] <= 40000
        pruned_before = st["pruned_subtrees"]
    assert pruned_before > 0, "the budget was never hit: the test does not exercise pruning"
    e.check_healthy()


def test_strict_mode_turns_pruning_and_adjudication_into_errors():
    """CCZ_FLAG_STRICT (round 6): where the throughput paths only COUNT a departure from the reference -- a kept subtree pruned to fit the
    node pool (the reference's tree is unbounded, mcts.py:31-39), a game adjudicated at max_plies (its game loop has no cap,
    game.py:155) -- the parity mode sets a sticky error bit and check_healthy() raises. Same workloads as the two counting tests."""
    from chinesechesszero_amd import _lib
    from chinesechesszero_amd._lib import CczError
    from chinesechesszero_amd.engine import SelfPlayEngine
    from chinesechesszero_amd.net import uniform_evaluator
    from test_gpu_soak import LinearEvaluator
    # (a) pruning: the workload of test_reroot_budget_prunes_deepest_levels_instead_of_failing
    B, n = 6, 300
    ev = LinearEvaluator(torch.device("cuda", 0), seed=2, sharp=14.0)
    e = SelfPlayEngine(B, n_playout=n, max_nodes=40000, reserve_nodes=37000, seed=3, strict=True)
    for move in range(6):
        leaf = e.select_leaves()
        for i in range(n):
            p, v = ev(leaf)
            leaf = e.step(p, v) if i + 1 < n else e.expand_backup(p, v)
        rc = e.root_children()
        forced = np.array([rc["acts"][b][int(np.argmax(rc["visits"][b][:rc["k"][b]]))] for b in range(B)], np.int32)
        e.finish_move(forced_moves=forced)
        if e.stats()["pruned_subtrees"]:
            break
    st = e.stats()
    assert st["pruned_subtrees"] > 0 and st["error_flags"] == _lib.ERR_PRUNED, st
    with pytest.raises(CczError, match="pruned"):
        e.check_healthy()
    # (b) adjudication at max_plies: 4 boards, cap 6 plies, one simulation per move
    e = SelfPlayEngine(4, n_playout=1, max_plies=6, seed=1, strict=True)
    for ply in range(7):
        leaf = e.select_leaves()
        e.expand_backup(*uniform_evaluator(leaf))
        e.finish_move()
    st = e.stats()
    assert st["truncated_games"] == 4 and st["error_flags"] == _lib.ERR_TRUNCATED, st
    with pytest.raises(CczError, match="max_plies"):
        e.check_healthy()
    # the same without the flag: counted, healthy
    e = SelfPlayEngine(4, n_playout=1, max_plies=6, seed=1)
    for ply in range(7):
        leaf = e.select_leaves()
        e.expand_backup(*uniform_evaluator(leaf))
        e.finish_move()
    assert e.stats()["truncated_games"] == 4 and e.stats()["error_flags"] == 0
    e.check_healthy()


@pytest.mark.parametrize("B,n", [(1, 1), (37, 1), (37, 2), (5, 3), (129, 7)])
def test_odd_sizes_and_minimal_playouts(B, n):
    """Board counts that are not multiples of anything and the degenerate n_playout = 1 (root expanded, no child
    visited: pi is uniform over the legal moves, mcts.py:165)."""
    from chinesechesszero_amd.selfplay import BatchedSelfPlay
    from chinesechesszero_amd.net import uniform_evaluator
    sp = BatchedSelfPlay(uniform_evaluator, B, n_playout=n, seed=B * 10 + n, max_plies=9)
    for move in range(12):
        rc_before = None
        if move == 0:
            leaf = sp.engine.select_leaves()
            for i in range(n):
                p, v = uniform_evaluator(leaf)
                if i + 1 < n:
                    leaf = sp.engine.step(p, v)
                else:
                    sp.engine.expand_backup(p, v)
            rc_before = sp.engine.root_children()
            assert np.all(rc_before["k"] == 44) and np.all(rc_before["root_visits"] == n)
            assert np.all(rc_before["visits"].sum(1) == n - 1)
            pi = sp.engine.root_pi(temps=1.0)
            if n == 1:
                assert np.allclose(pi[:, :44], 1.0 / 44, atol=1e-12)
            moves = sp.finish_move().cpu().numpy()
        else:
            moves = sp.run_move().cpu().numpy()
        st = sp.engine.game_status()
        assert np.all((moves >= 0) | (st["plies"] >= 9) | (st["over"] == 1))
        if st["over"].any():
            s, p, z = sp.harvest()
            assert s.shape[0] == 2 * int(st["plies"][st["over"] == 1].sum())
    sp.engine.check_healthy()
    assert sp.engine.stats()["games"] >= B
