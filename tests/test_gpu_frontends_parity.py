"""GPU: the SURVEY 8f rows compared WITH THE ORACLE (round 1 had property tests only): batched evaluation matches,
the UCI loop's bestmove, model hot-reload (eager and through a captured hipGraph)."""
import io

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_batched_match_every_move_is_the_sequential_searchs_choice_and_games_replay_on_the_oracle():
    """match.BatchedMatch == Game.start_play with two non-self-play MCTS_AI players (game.py:77-130, mcts.py:225-229).
    A sample of boards is mirrored on sequential oracle searches fed with the same evaluator numbers, tree discarded after
    every move: root children N / Q / P bit-exact before EVERY move, and the move played is an arg-max-visit child (what
    temperature 1e-3 selects up to exact ties). Every game of the batch is then replayed on the oracle's rules."""
    from gpu_harness import SampleMirror
    from oracle import OracleBoard
    from test_gpu_soak import LinearEvaluator
    from chinesechesszero_amd.match import BatchedMatch
    B, n, cap = 48, 24, 90
    dev = torch.device("cuda", 0)
    red, black = LinearEvaluator(dev, seed=1, sharp=8.0), LinearEvaluator(dev, seed=2, sharp=3.0)
    m = BatchedMatch(None, None, B, n_playout=n, seed=5, max_plies=cap)
    sample = list(range(0, B, 4))
    sm = SampleMirror(m.engine, sample, check_every=1)
    games = [[] for _ in range(B)]

    def wrap(ev):
        def f(leaf):
            prob, value = ev(leaf)
            over = m.engine.game_status()["over"]
            live = [j for j, b in enumerate(sm.sample) if not over[b]]
            keep = (sm.sample, sm.boards, sm.mcts, sm.idx)
            # finished boards idle (no pending leaf): mirror the live ones only
            sm.sample, sm.boards, sm.mcts = [keep[0][j] for j in live], [keep[1][j] for j in live], [keep[2][j] for j in live]
            sm.idx = keep[3][live]
            try:
                if live:
                    sm.backup_on_oracles(prob, value)
            finally:
                sm.sample, sm.boards, sm.mcts, sm.idx = keep
            return prob, value
        return f

    m.ev = {1: wrap(red), 0: wrap(black)}
    seen = {"moves": 0}

    def before_move(match):
        rc = match.engine.root_children()
        over = match.engine.game_status()["over"]
        match._rc, match._over = rc, over
        for j, b in enumerate(sm.sample):
            if over[b]:
                continue
            acts, visits, q, prior = sm.mcts[j].root_children()
            k = len(acts)
            assert rc["k"][b] == k and np.array_equal(rc["acts"][b][:k], acts.astype(np.uint16))
            assert np.array_equal(rc["visits"][b][:k], visits), (b, rc["visits"][b][:k], visits)
            assert np.array_equal(rc["q"][b][:k].view(np.uint32), q.view(np.uint32))
            assert rc["root_visits"][b] == n     # a fresh tree every move (mcts.py:228-229)

    def on_move(match, moves):
        rc, over = match._rc, match._over
        now = match.engine.game_status()
        for b in range(B):
            if over[b]:
                assert moves[b] == -1
                continue
            if moves[b] < 0:   # the documented ply cap: the game is adjudicated INSTEAD of moving
                assert now["over"][b] and now["plies"][b] == cap and now["winner"][b] == -1
                continue
            k = rc["k"][b]
            v = rc["visits"][b][:k]
            assert v[list(rc["acts"][b][:k]).index(moves[b])] == v.max(), (b, moves[b], v)   # arg-max visits (temp 1e-3)
            games[b].append(int(moves[b]))
            seen["moves"] += 1
        for j, b in enumerate(sm.sample):
            if not over[b] and moves[b] >= 0:
                sm.mcts[j].update_with_move(-1)
                sm.boards[j].push_id(int(moves[b]))

    res = m.play(before_move=before_move, on_move=on_move)
    assert res["unfinished"] == 0 and res["red_wins"] + res["black_wins"] + res["draws"] == B
    st = m.engine.game_status()
    decided = 0
    for b in range(B):
        ob = OracleBoard()
        for t, mv in enumerate(games[b]):
            assert not ob.is_game_over() and mv in ob.legal_ids(), (b, t, mv)
            ob.push_id(mv)
        assert len(games[b]) == st["plies"][b]
        if ob.is_game_over():
            o = ob.outcome()
            assert int(st["winner"][b]) == (-1 if o.winner is None else int(o.winner)), b
            decided += int(o.winner is not None)
        else:
            assert st["plies"][b] == cap and st["winner"][b] == -1        # adjudicated at the documented cap
    assert seen["moves"] == int(st["plies"].sum()) == m.engine.stats()["moves"]
    assert res["red_wins"] + res["black_wins"] == decided
    print("match:", res["red_wins"], res["black_wins"], res["draws"], "plies", int(st["plies"].sum()))


def _hash_policy(salt, scale=40.0):
    from oracle.evaluators import hash_eval

    def policy(board, red_states=None, black_states=None):
        ids = board.legal_ids()
        p, v = hash_eval(board.squares()[None, :], np.array([1 if board.turn else 0]), salt=salt, scale=scale)
        return zip(ids, p[0][ids]), np.array([[v[0]]], dtype=np.float32)

    return policy


def _oracle_bestmove(board, salt, nodes, seed):
    """What the reference's non-self-play get_action returns (mcts.py:214,225-229) on a sequential search."""
    from oracle import OracleMCTS
    from oracle.evaluators import hash_eval

    def ev(b, ids):
        p, v = hash_eval(b.squares()[None, :], np.array([1 if b.turn else 0]), salt=salt, scale=40.0)
        return p[0][ids], v[0]

    mcts = OracleMCTS(ev, c_puct=5, n_playout=nodes)
    acts, visits, _ = mcts.get_move_probs(board, 1e-3)
    x = 1.0 / 1e-3 * np.log(visits.astype(np.int64) + 1e-10)      # tools.softmax, mcts.py:165
    probs = np.exp(x - np.max(x))
    probs /= probs.sum()
    np.random.seed(seed)
    return int(np.random.choice(acts, p=probs)), acts, visits


def test_uci_go_returns_exactly_the_sequential_searchs_move():
    """`go nodes N` == the move a sequential reference-style search picks under the same np.random state, over a game
    that includes captures, a repetition history and an opponent's replies the tree never saw (position ... moves ...)."""
    import oracle
    from oracle import OracleBoard
    from chinesechesszero_amd.uci import UciLoop
    L = oracle.lib()
    out = io.StringIO()
    loop = UciLoop(policy_value_fn=_hash_policy(9), n_playout=64, out=out)
    assert loop.handle("uci") and loop.handle("ucinewgame")
    moves = ["b0c2", "b9c7", "c2b0", "c7b9", "b0c2", "b9c7", "c2b0", "c7b9", "h2e2"]   # start position already seen three times
    ob = OracleBoard()
    for u in moves:
        ob.push(u)
    played = list(moves)
    for rnd, nodes in enumerate((48, 80, 33, 64, 64)):
        seed = 100 + rnd
        np.random.seed(seed)
        loop.handle("position startpos moves " + " ".join(played))
        loop.handle(f"go nodes {nodes}")
        best = [l for l in out.getvalue().splitlines() if l.startswith("bestmove")][-1].split()[1]
        want, acts, visits = _oracle_bestmove(ob, 9, nodes, seed)
        assert best == L.xq_move_uci(want).decode(), (rnd, best, L.xq_move_uci(want).decode())
        info = [l for l in out.getvalue().splitlines() if l.startswith("info nodes")][-1]
        assert info.startswith(f"info nodes {nodes} ") and f"visits {int(visits.max())}/{int(visits.sum())}" in info
        ob.push(best)
        played.append(best)
        reply = L.xq_move_uci(ob.legal_ids()[(7 * rnd + 3) % len(ob.legal_ids())]).decode()   # the opponent's move
        ob.push(reply)
        played.append(reply)
    # a FEN start with a clock: sixty-move / repetition state comes from the FEN + moves, not from the opening
    fen = "3k5/9/9/9/9/9/9/9/1R7/R2K5 w - - 117 80"
    loop.handle("ucinewgame")
    np.random.seed(7)
    loop.handle(f"position fen {fen}")
    loop.handle("go nodes 40")
    best = [l for l in out.getvalue().splitlines() if l.startswith("bestmove")][-1].split()[1]
    from golden_cases import sq
    sqs = np.zeros(90, np.uint8)
    sqs[sq("d0")], sqs[sq("a0")], sqs[sq("b1")], sqs[sq("d9")] = 7, 3, 3, 15
    want, _, _ = _oracle_bestmove(OracleBoard.from_array(sqs, 1, 117), 9, 40, 7)
    assert best == L.xq_move_uci(want).decode()


def test_model_hot_reload_changes_the_evaluator_eager_and_graphed():
    """broadcast_model (group of one: the reload + refresh half of it) makes the evaluator answer with the new weights;
    a search that replays a captured hipGraph must follow -- the graph holds device addresses of the OLD inference copy
    (stale or freed weights were replayed silently)."""
    from chinesechesszero_amd.net import PolicyValueNet
    from chinesechesszero_amd.replay import broadcast_model
    from chinesechesszero_amd.selfplay import BatchedSelfPlay
    dev = "cuda:0"
    torch.manual_seed(1)
    a = PolicyValueNet(device=dev, num_channels=32, resblocks_num=2)
    torch.manual_seed(2)
    b = PolicyValueNet(device=dev, num_channels=32, resblocks_num=2)
    B, n = 8, 12
    graphed = BatchedSelfPlay(a.evaluate_leaves_logits, B, n_playout=n, seed=3, use_graph=True)
    eager = BatchedSelfPlay(a.evaluate_leaves_logits, B, n_playout=n, seed=3, use_graph=False)

    def same_trees():
        r1, r2 = graphed.engine.root_children(), eager.engine.root_children()
        return all(np.array_equal(r1[k], r2[k]) for k in ("k", "acts", "visits", "root_visits")) and \
            np.array_equal(r1["q"].view(np.uint32), r2["q"].view(np.uint32)) and np.array_equal(r1["prior"].view(np.uint32), r2["prior"].view(np.uint32))

    for sp in (graphed, eager):
        sp.run_move()
    assert same_trees() and graphed._graph.captures == 1
    leaf = eager.engine.select_leaves().clone()
    before = a.evaluate_leaves_logits(leaf)[0].float().clone()
    # the trainer's rank publishes new weights: here they arrive by load_state_dict, broadcast_model then refreshes
    v0 = a.weights_version
    a.policy_value_net.load_state_dict(b.policy_value_net.state_dict())
    broadcast_model(a, src=0)
    assert a.weights_version > v0
    after = a.evaluate_leaves_logits(leaf)[0].float()
    assert float((before - after).abs().max()) > 1e-2             # other weights, other logits
    assert torch.equal(after, b.evaluate_leaves_logits(leaf)[0].float())   # the evaluator now IS the source rank's
    for sp in (graphed, eager):
        sp.run_move()
    assert same_trees() and graphed._graph.captures == 2           # re-captured against the new inference copy
    # a training step invalidates the copy too
    rows = 64
    st = (torch.rand((rows, 17, 7, 10, 9), device=dev) > 0.9).float()
    pi = torch.softmax(torch.randn(rows, 2086, device=dev), 1)
    a.train_step(st, pi, torch.zeros(rows, device=dev), lr=1e-3)
    for sp in (graphed, eager):
        sp.run_move()
    assert same_trees() and graphed._graph.captures == 3
    graphed.engine.check_healthy()
