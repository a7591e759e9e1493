"""GPU: long lockstep self-play with staggered game ends; every finished game is replayed on the oracle:
all moves legal, game end and winner agree, harvested rows consistent. Also the full-size (4096-board)
invariants of the tree arrays."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class LinearEvaluator:
    """A tiny deterministic device-side 'net': softmax / tanh of a fixed random projection of the live planes."""

    batched = True

    def __init__(self, device, seed=0, sharp=6.0):
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.W = (torch.randn(1890, 2086, generator=g) * sharp / 5.6).to(device)
        self.w = (torch.randn(1890, generator=g) * 0.7).to(device)

    def __call__(self, leaf):
        B = leaf.shape[0]
        x = leaf.view(B, 17, 630)
        x = torch.cat([x[:, 7], x[:, 15], x[:, 16]], dim=1).float()
        return torch.softmax(x @ self.W, dim=1).contiguous(), torch.tanh(x @ self.w).contiguous()


@pytest.mark.parametrize("mode", ["dense_eager", "net_logits_hipgraph", "net_planned_cache_verify"])
def test_full_games_replay_on_oracle(mode):
    from oracle import OracleBoard
    from chinesechesszero_amd.selfplay import BatchedSelfPlay
    B, n, max_plies, n_moves = 96, 10, 160, 420
    if mode == "dense_eager":
        ev = LinearEvaluator(torch.device("cuda", 0))
        sp = BatchedSelfPlay(ev, B, n_playout=n, seed=42, max_plies=max_plies)
    elif mode == "net_logits_hipgraph":   # the real evaluator boundary: PyTorch net -> logits -> ccz_gather_priors, replayed as a hipGraph
        from chinesechesszero_amd.net import PolicyValueNet
        torch.manual_seed(5)
        pvn = PolicyValueNet(device="cuda:0", num_channels=32, resblocks_num=2)
        B, n_moves = 64, 260
        sp = BatchedSelfPlay(pvn.evaluate_leaves_logits, B, n_playout=n, seed=43, max_plies=max_plies, use_graph=True)
    else:   # the planned boundary on the hand-written tower kernels (256 wide) with the evaluation cache in its verify mode: whole
        # games, restarts, thousands of table hits of which ~1 % are evaluated again and must come back bit-identical
        from chinesechesszero_amd.net import PolicyValueNet
        torch.manual_seed(6)
        pvn = PolicyValueNet(device="cuda:0", num_channels=256, resblocks_num=1)
        B, n_moves = 80, 260
        sp = BatchedSelfPlay(pvn.evaluate_leaves_logits, B, n_playout=n, seed=44, max_plies=max_plies, eval_cache_log2=16, cache_verify=True)
        assert sp.planned
    e = sp.engine
    games = [[] for _ in range(B)]
    finished = decisive = natural_draws = truncated = rows_total = 0
    for mv in range(n_moves):
        moves = sp.run_move().cpu().numpy()
        st = e.game_status()
        for b in range(B):
            if moves[b] >= 0:
                games[b].append(int(moves[b]))
        if st["over"].any():
            idx = np.nonzero(st["over"])[0]
            states, pi, z = sp.harvest()
            states, pi, z = states.cpu().numpy(), pi.cpu().numpy(), z.cpu().numpy()
            off = 0
            for b in idx:
                T = int(st["plies"][b])
                assert T == len(games[b]) or (T == max_plies and len(games[b]) == max_plies)
                ob = OracleBoard()
                for t, m in enumerate(games[b]):
                    assert m in ob.legal_ids(), (b, t, m)
                    assert not ob.is_game_over()
                    # the recorded state's newest history slot is the position before the move
                    red, black = ob.decode()
                    assert np.array_equal(states[off + t][0], red) and np.array_equal(states[off + t][8], black)
                    assert np.all(states[off + t][16] == (1 if ob.turn else 0))
                    assert pi[off + t][m] > 0 and abs(pi[off + t].sum() - 1) < 1e-5
                    assert np.count_nonzero(pi[off + t]) <= len(ob.legal_ids())
                    ob.push_id(m)
                w = int(st["winner"][b])
                if ob.is_game_over():
                    o = ob.outcome()
                    want = -1 if o.winner is None else (1 if o.winner else 0)
                    assert w == want, (b, w, want)
                    if want >= 0:
                        decisive += 1
                    else:
                        natural_draws += 1
                else:
                    assert T == max_plies and w == -1  # adjudicated at the documented cap
                    truncated += 1
                zs = z[off:off + T]
                turns = np.array([1 if (t % 2 == 0) else 0 for t in range(T)])
                assert np.array_equal(zs, np.zeros(T) if w < 0 else np.where(turns == w, 1.0, -1.0))
                assert np.array_equal(z[off + T:off + 2 * T], zs)  # mirrored half carries the same z
                off += 2 * T
                finished += 1
                games[b] = []
            assert off == states.shape[0]
            rows_total += off
    s = e.stats()
    e.check_healthy()
    assert s["games"] == finished and s["truncated_games"] == truncated
    if mode == "net_planned_cache_verify":
        assert s["cache_hits"] > 1000 and s["cache_verified"] > 10 and s["cache_verify_mismatches"] == 0, s
    assert finished >= B and (decisive + natural_draws > 0 or mode != "dense_eager"), (finished, decisive, natural_draws, truncated)
    print("soak:", dict(finished=finished, decisive=decisive, draws=natural_draws, truncated=truncated, rows=rows_total,
                        depth_peak=s["depth_peak"], nodes_peak=s["nodes_peak"]))


def test_tree_invariants_at_full_size():
    """4096 boards x 64 sims: size-independent properties of the device tree."""
    from chinesechesszero_amd.selfplay import BatchedSelfPlay
    B, n = 4096, 64
    ev = LinearEvaluator(torch.device("cuda", 0), seed=3, sharp=10.0)
    sp = BatchedSelfPlay(ev, B, n_playout=n, seed=7)
    e = sp.engine
    carried = np.zeros(B, np.int64)
    for ply in range(3):
        leaf = e.select_leaves()
        for i in range(n):
            p, v = ev(leaf)
            if i + 1 < n:
                leaf = e.step(p, v)
            else:
                e.expand_backup(p, v)
        rc = e.root_children()
        k = rc["k"]
        assert np.all(k > 0) and np.all(k <= 128)
        tot = np.array([rc["visits"][b][:k[b]].sum() for b in range(B)])
        # every playout visits the root; all but the one that expanded the root descend into a child
        assert np.all(rc["root_visits"] == carried + n)
        assert np.all(tot == rc["root_visits"] - 1)
        acts = rc["acts"]
        assert all(np.all(np.diff(acts[b][:k[b]].astype(int)) > 0) for b in range(0, B, 97))  # ascending ids
        q = rc["q"]
        assert np.all(np.abs(q) <= 1.0 + 1e-6)
        pi = e.root_pi(temps=1.0)
        assert np.allclose(pi.sum(1), 1.0, atol=1e-12)
        moves = sp.finish_move().cpu().numpy()
        chosen = np.array([rc["visits"][b][list(rc["acts"][b][:k[b]]).index(moves[b])] for b in range(B)])
        carried = chosen  # tree reuse: the new root keeps the chosen child's visits (mcts.py:172-175)
        assert np.array_equal(e.root_children()["root_visits"], carried)
    st = e.stats()
    assert st["sims"] == 3 * B * n and st["moves"] == 3 * B
    e.check_healthy()
