"""GPU: per-move bookkeeping and training-tuple harvest against a NumPy restatement of
reference game.py:133-237 + collect.py:64-131 built on the oracle's positions."""
import numpy as np
import pytest
import torch

from golden_cases import STARTS

pytestmark = pytest.mark.gpu


def _expected_rows(positions, turns, pis, winner, fm, quirks, mirror):
    """positions[t] = squares before move t; pis[t] dense float64[2086]; returns (states, pi, z)."""
    T = len(positions)

    def planes(sq):
        red = np.zeros((7, 90), np.float16)
        black = np.zeros((7, 90), np.float16)
        for s in np.nonzero(sq)[0]:
            pc = int(sq[s])
            (black if pc & 8 else red)[(pc & 7) - 1, s] = 1
        return red.reshape(7, 10, 9), black.reshape(7, 10, 9)

    dec = [planes(p) for p in positions]
    rows_s, rows_p, rows_z = [], [], []
    for t in range(T):
        te = T - 1 if quirks else t
        red = [dec[max(te - i, 0)][0] for i in range(8)]      # game.py:36-44 newest first, start position before that
        black = [dec[max(te - i, 0)][1] for i in range(8)]
        turn_plane = np.ones((1, 7, 10, 9), np.float16) if (quirks or turns[t]) else np.zeros((1, 7, 10, 9), np.float16)
        rows_s.append(np.concatenate((red, black, turn_plane), axis=0))
        rows_p.append(pis[t])
        rows_z.append(0.0 if winner < 0 else (1.0 if turns[t] == winner else -1.0))
    if mirror:
        n = len(rows_s)
        for i in range(n):
            rows_s.append(np.stack([np.flip(g, axis=2) for g in rows_s[i]]))
            rows_p.append(rows_p[i][fm])
            rows_z.append(rows_z[i])
    return np.stack(rows_s), np.stack(rows_p), np.array(rows_z, np.float32)


@pytest.mark.parametrize("quirks,mirror", [(False, True), (True, True), (False, False)])
def test_harvest_matches_numpy_restatement(quirks, mirror):
    import oracle
    from oracle import OracleBoard
    from chinesechesszero_amd.engine import SelfPlayEngine
    from chinesechesszero_amd.net import uniform_evaluator
    fm = oracle.flip_map()
    B, n, max_plies = 4, 6, 12
    e = SelfPlayEngine(B, n_playout=n, max_plies=max_plies, reference_quirks=quirks, mirror=mirror, seed=5)
    boards = [OracleBoard(),
              OracleBoard.from_array(STARTS["two_rooks"], 1, 0),
              OracleBoard.from_array(STARTS["capture_to_bare"], 1, 3),
              OracleBoard.from_array(STARTS["rook_knight"], 0, 100)]
    for b in range(1, B):
        e.set_position(b, boards[b].squares(), 1 if boards[b].turn else 0, boards[b].halfmove)
    scripted = {1: ["a7a9"], 2: ["e0e1"]}
    rs = np.random.RandomState(11)
    hist = [dict(pos=[], turn=[], pi=[]) for _ in range(B)]
    done = [False] * B
    winner = [-1] * B
    for ply in range(max_plies + 1):
        for _ in range(n):
            leaf = e.select_leaves()
            p, v = uniform_evaluator(leaf)
            e.expand_backup(p, v)
        rc = e.root_children()
        temps = np.array([1.0 if ply + 1 <= 30 else 0.5] * B)
        pi = e.root_pi(temps=temps)
        forced = np.full(B, -1, np.int32)
        for b in range(B):
            if done[b]:
                continue
            if len(hist[b]["pos"]) >= max_plies:  # the engine adjudicates a draw at the cap
                done[b] = True
                continue
            ids = boards[b].legal_ids()
            k = int(rc["k"][b])
            assert rc["acts"][b][:k].tolist() == ids
            if b in scripted and ply < len(scripted[b]):
                mv = oracle.lib().xq_move_id(*[(ord(s[0]) - 97) + 9 * int(s[1]) for s in (scripted[b][ply][:2], scripted[b][ply][2:])])
            else:
                mv = ids[rs.randint(len(ids))]
            forced[b] = mv
            dense = np.zeros(2086)
            dense[ids] = pi[b][:k]
            hist[b]["pos"].append(boards[b].squares())
            hist[b]["turn"].append(1 if boards[b].turn else 0)
            hist[b]["pi"].append(dense)
            boards[b].push_id(mv)
            if boards[b].is_game_over() or boards[b].is_tie():
                done[b] = True
                o = boards[b].outcome()
                winner[b] = -1 if (o is None or o.winner is None) else (1 if o.winner else 0)
        moves = e.finish_move(forced_moves=forced, temps=temps).cpu().numpy()
        assert all(moves[b] == forced[b] for b in range(B) if forced[b] >= 0)
        if all(done):
            break
    e.finish_move(forced_moves=np.full(B, -1, np.int32))  # lets boards at the cap adjudicate
    st = e.game_status()
    assert st["over"].tolist() == [1] * B
    assert st["winner"].tolist() == winner
    assert winner[1] == 1 and winner[2] == -1
    states, pis, z = e.harvest()
    exp = [_expected_rows(h["pos"], h["turn"], h["pi"], w, fm, quirks, mirror) for h, w in zip(hist, winner)]
    S = np.concatenate([x[0] for x in exp])
    P = np.concatenate([x[1] for x in exp])
    Z = np.concatenate([x[2] for x in exp])
    assert states.shape[0] == S.shape[0] == sum(len(h["pos"]) for h in hist) * (2 if mirror else 1)
    assert np.array_equal(states.cpu().numpy(), S)
    assert np.allclose(pis.cpu().numpy(), P, rtol=0, atol=1e-7)      # float32 rows of float64 pi
    assert np.array_equal(z.cpu().numpy(), Z)
    assert np.allclose(pis.sum(1).cpu().numpy(), 1.0, atol=1e-5)
    # harvested boards restarted from the opening position with fresh trees
    st = e.game_status()
    assert st["over"].sum() == 0 and st["plies"].sum() == 0
    from golden_cases import start_position
    assert np.array_equal(e.root_positions(), np.stack([start_position()] * B))
    assert e.root_children()["k"].sum() == 0
    e.check_healthy()


def test_staggered_restart_and_stats():
    """Boards finish at different plies, are harvested and restarted while the others keep searching."""
    from chinesechesszero_amd.selfplay import BatchedSelfPlay
    from chinesechesszero_amd.net import uniform_evaluator
    sp = BatchedSelfPlay(uniform_evaluator, 32, n_playout=8, seed=1, max_plies=10)
    rows = 0
    for move in range(25):
        sp.run_move()
        st = sp.engine.game_status()
        if st["over"].any():
            s, p, z = sp.harvest()
            assert s.shape[0] == 2 * int(st["plies"][st["over"] == 1].sum())
            rows += s.shape[0]
    stats = sp.engine.stats()
    assert stats["sims"] == 32 * 8 * 25
    assert stats["games"] >= 32 * 2 and rows >= 32 * 2 * 10 * 2
    assert stats["truncated_games"] == stats["games"]
    sp.engine.check_healthy()


def test_harvest_in_chunks_bounded_by_capacity():
    """All boards reach the ply cap in the same move: the rows come out in capacity-bounded chunks, nothing lost."""
    from chinesechesszero_amd._lib import CczError
    from chinesechesszero_amd.selfplay import BatchedSelfPlay
    from chinesechesszero_amd.net import uniform_evaluator
    B, cap = 24, 6
    sp = BatchedSelfPlay(uniform_evaluator, B, n_playout=4, seed=9, max_plies=cap)
    for _ in range(cap + 1):
        sp.run_move()
    st = sp.engine.game_status()
    assert st["over"].all() and (st["plies"] == cap).all()
    with pytest.raises(CczError, match="harvest_chunks"):
        sp.engine.harvest(max_rows=40)
    total = 0
    sizes = []
    for s, p, z in sp.harvest_chunks(max_rows=40):      # 12 rows per game -> 3 games per chunk
        assert s.shape[0] <= 40 and s.shape[0] % (2 * cap) == 0
        assert torch.allclose(p.sum(1), torch.ones_like(p[:, 0]), atol=1e-5)
        sizes.append(s.shape[0])
        total += s.shape[0]
    assert total == B * cap * 2 and sizes == [36] * 8
    st = sp.engine.game_status()
    assert st["over"].sum() == 0 and st["plies"].sum() == 0
    # a single game longer than the chunk still comes out (the buffer grows to fit it)
    for _ in range(cap + 1):
        sp.run_move()
    got = sum(s.shape[0] for s, _, _ in sp.harvest_chunks(max_rows=5))
    assert got == B * cap * 2
    sp.engine.check_healthy()


def _play_out(B, seed, quirks, mirror, plane_of_type=None):
    """Self-play with the stub evaluator until every board has finished at least one game; games end at different plies
    (captures to bare kings, repetition, the 14-ply cap)."""
    from chinesechesszero_amd.selfplay import BatchedSelfPlay
    from chinesechesszero_amd.net import uniform_evaluator
    sp = BatchedSelfPlay(uniform_evaluator, B, n_playout=4, seed=seed, max_plies=14, reference_quirks=quirks, mirror=mirror,
                         plane_of_type=plane_of_type)
    e = sp.engine
    e.set_position(1, STARTS["capture_to_bare"], 1, 3)
    e.set_position(2, STARTS["two_rooks"], 1, 0)
    e.set_position(3, STARTS["rook_knight"], 0, 100)
    for _ in range(16):
        sp.run_move()
    assert sp.engine.game_status()["over"].all()
    return sp


@pytest.mark.parametrize("quirks,mirror,pot", [(False, True, None), (True, True, None), (False, False, (0, 6, 5, 4, 3, 2, 1, 0))])
def test_compact_records_expand_to_the_dense_harvest_byte_for_byte(quirks, mirror, pot):
    """The multi-GPU wire format: ccz_harvest_records + ccz_expand_records == ccz_harvest, bit for bit, for the same games
    (two engines with the same seed play the same games); headers agree with the game status."""
    from chinesechesszero_amd.engine import expand_records, game_aligned_chunks
    B = 12
    a, b = _play_out(B, 21, quirks, mirror, pot), _play_out(B, 21, quirks, mirror, pot)
    st = a.engine.game_status()
    S, P, Z = a.engine.harvest()
    recs = list(b.engine.harvest_record_chunks(40))          # several chunks, whole games each
    assert len(recs) > 1 and all(r.shape[0] <= 40 or r.shape[0] == int(r[0, 98:100].view(torch.int16)) for r in recs)
    rec = torch.cat(recs)
    mul = 2 if mirror else 1
    assert rec.shape == (int(st["plies"].sum()), 880) and S.shape[0] == mul * rec.shape[0]
    hdr = rec[:, 96:112].cpu().numpy()
    t = hdr[:, 0:2].copy().view(np.uint16).ravel()
    T = hdr[:, 2:4].copy().view(np.uint16).ravel()
    first = np.nonzero(t == 0)[0]
    assert T[first].tolist() == st["plies"].tolist() and hdr[first, 4].view(np.int8).tolist() == st["winner"].tolist()
    assert hdr[:, 8:12].copy().view(np.uint32).ravel()[first].tolist() == list(range(B))       # global board ids, board order
    assert (rec[:, 90:96] == 0).all() and (hdr[:, 7] == 0).all()
    flags = b.engine.record_flags()
    # whole buffer at once, and chunk by chunk: the same rows
    s1, p1, z1 = expand_records(rec, flags, pot)
    assert torch.equal(s1, S) and torch.equal(p1, P) and torch.equal(z1, Z)
    parts = [expand_records(c.contiguous(), flags, pot) for c in game_aligned_chunks(rec, 25)]
    assert len(parts) > 2 and all(torch.equal(torch.cat([q[i] for q in parts]), (S, P, Z)[i]) for i in range(3))
    st2 = b.engine.game_status()
    assert st2["over"].sum() == 0 and st2["plies"].sum() == 0       # harvested boards restarted
    a.engine.check_healthy()
    b.engine.check_healthy()


def test_records_expand_into_a_replay_ring_and_cut_games_are_refused():
    from chinesechesszero_amd.engine import expand_records
    from chinesechesszero_amd.replay import ReplayBuffer
    a, b = _play_out(8, 4, False, True), _play_out(8, 4, False, True)
    S, P, Z = a.engine.harvest()
    rec = torch.cat(list(b.engine.harvest_record_chunks(1 << 16)))
    R = S.shape[0]
    rb = ReplayBuffer(R + 10, "cuda")          # the ring wraps: the rows land at (head + i) % capacity
    rb.head = R - 3
    bad = torch.zeros(1, dtype=torch.int32, device="cuda")
    assert rb.append_records(rec, 0, None, bad=bad) == R and rb.head == (R - 3 + R) % (R + 10) and int(bad.item()) == 0
    idx = (torch.arange(R, device="cuda") + (R - 3)) % (R + 10)
    assert torch.equal(rb.states[idx], S) and torch.equal(rb.pi[idx], P) and torch.equal(rb.z[idx], Z)
    small = ReplayBuffer(64, "cuda")           # smaller than one exchange: game by game, the newest rows survive
    assert small.append_records(rec, 0) == R and small.total == R and small.size == 64
    # a buffer that starts in the middle of a game: those records are skipped and counted, nothing is read out of bounds
    T0 = int(rec[0, 98:100].view(torch.int16))
    s, p, z = expand_records(rec[1:].contiguous(), 0, None, bad=bad)
    torch.cuda.synchronize()
    assert int(bad.item()) == T0 - 1
    assert torch.equal(s[2 * (T0 - 1):], S[2 * T0:]) and torch.equal(z[2 * (T0 - 1):], Z[2 * T0:])
    with pytest.raises(ValueError):
        expand_records(rec.view(-1)[:-1], 0)


class _FakeH5:
    """A recording stand-in for the h5py calls of reference collect.py:146-167 (File(path, "a") as context manager, attrs,
    create_group, create_dataset): h5py itself is not in this image."""
    files = {}

    class _Group(dict):
        def create_dataset(self, name, data=None, compression=None):
            self[name] = (np.asarray(data), compression)

    class File:
        def __init__(self, path, mode):
            assert mode == "a"
            self.store = _FakeH5.files.setdefault(path, {"attrs": {}, "groups": {}})
            self.attrs = self.store["attrs"]

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def create_group(self, name):
            assert name not in self.store["groups"]
            g = _FakeH5._Group()
            self.store["groups"][name] = g
            return g


def test_per_game_hdf5_groups_in_the_references_layout():
    """collect.py:146-167: group game_{i} per game, states / mcts_probs (gzip) / winners, attrs["iters"]; rows = the dense
    harvest, game by game (samples then mirror images)."""
    from chinesechesszero_amd.collect import write_games_hdf5
    _FakeH5.files.clear()
    a, b = _play_out(6, 9, False, True), _play_out(6, 9, False, True)
    st = a.engine.game_status()
    S, P, Z = (t.cpu().numpy() for t in a.engine.harvest())
    rec = torch.cat(list(b.engine.harvest_record_chunks(1 << 16)))
    assert write_games_hdf5(rec[:int(st["plies"][:2].sum())], "mem.h5", 0, None, h5py_module=_FakeH5) == 2   # two calls append
    assert write_games_hdf5(rec[int(st["plies"][:2].sum()):], "mem.h5", 0, None, h5py_module=_FakeH5) == 6
    f = _FakeH5.files["mem.h5"]
    assert f["attrs"]["iters"] == 6 and sorted(f["groups"]) == [f"game_{i}" for i in range(6)]
    lo = 0
    for i in range(6):
        g = f["groups"][f"game_{i}"]
        T2 = 2 * int(st["plies"][i])
        assert g["states"][1] == "gzip" and g["mcts_probs"][1] == "gzip" and g["winners"][1] is None
        assert g["states"][0].dtype == np.float16 and g["mcts_probs"][0].dtype == np.float64
        assert np.array_equal(g["states"][0], S[lo:lo + T2]) and np.array_equal(g["mcts_probs"][0], P[lo:lo + T2].astype(np.float64))
        assert np.array_equal(g["winners"][0], Z[lo:lo + T2].astype(np.float64))
        lo += T2
    assert lo == S.shape[0]
    with pytest.raises(ImportError, match="h5py"):
        write_games_hdf5(rec, "x.h5")
