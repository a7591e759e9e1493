"""CPU: the N>1 exchange path (all-gather of finished tuples) under gloo, world_size 2."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rows(rank, n):
    g = torch.Generator().manual_seed(100 + rank)
    s = (torch.rand((n, 17, 7, 10, 9), generator=g) > 0.9).to(torch.float16)
    p = torch.rand((n, 2086), generator=g)
    z = torch.full((n,), float(rank + 1))
    return s, p, z


def _worker(rank, world, port, counts_per_round, cap, q):
    try:
        _worker_body(rank, world, port, counts_per_round, cap, q)
    except Exception as e:  # surface the failure instead of letting the parent wait for its timeout
        q.put((rank, False, repr(e)))


def _worker_body(rank, world, port, counts_per_round, cap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chinesechesszero_amd.replay import ReplayBuffer, TupleGatherer
    g = TupleGatherer(cap, "cpu")
    rb = ReplayBuffer(64, "cpu")
    ok = True
    for counts in counts_per_round:
        s, p, z = _rows(rank, counts[rank])
        S, P, Z = g.gather(s, p, z, more=(rank == 1 and counts[1] == 2), user=rank + 1)
        # ONE collective per exchange unless some rank holds more rows than the buffer; counts, the "more" flag and the
        # user counter travel in the header of that same collective
        ok = ok and g.collectives == max(1, -(-max(counts) // cap)) and g.rows_per_rank == list(counts)
        ok = ok and g.user_sum == 3 and g.any_more == (counts[1] == 2)
        exp = [_rows(r, counts[r]) for r in range(world)]
        ES = torch.cat([e[0] for e in exp])
        EP = torch.cat([e[1] for e in exp])
        EZ = torch.cat([e[2] for e in exp])
        ok = ok and S.shape[0] == sum(counts) and torch.equal(S, ES) and torch.equal(P, EP) and torch.equal(Z, EZ)
        rb.append(S, P, Z)
    q.put((rank, ok, rb.total))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_of_tuples_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    # uneven counts, an empty rank, and a round that needs two passes through the padded buffer (cap 4)
    rounds = [(3, 1), (0, 2), (0, 0), (9, 4)]
    procs = [ctx.Process(target=_worker, args=(r, world, port, rounds, 4, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert [r[1] for r in res] == [True, True]
    assert res[0][2] == res[1][2] == sum(sum(c) for c in rounds)  # every rank holds the union of the shards


def _records(rank, game_lengths):
    """Synthetic compact ply records (include/cczero.h CCZ_REC_*): whole games, valid headers, random payload."""
    g = torch.Generator().manual_seed(500 + rank)
    P = sum(game_lengths)
    rec = torch.randint(0, 256, (P, 880), dtype=torch.uint8, generator=g)
    hdr = np.zeros((P, 4), np.uint16)
    p = 0
    for T in game_lengths:
        for t in range(T):
            hdr[p] = (t, T, 0, 0)
            p += 1
    rec[:, 96:104] = torch.from_numpy(hdr.view(np.uint8).reshape(P, 8))
    return rec


def _chunks(rank, lengths, max_plies):
    """What an engine's harvest_record_chunks(max_plies) hands over: chunks of whole games, in order."""
    rec, out, lo = _records(rank, lengths), [], 0
    bounds = np.cumsum([0] + list(lengths))
    while lo < len(lengths):
        hi = lo + 1
        while hi < len(lengths) and bounds[hi + 1] - bounds[lo] <= max_plies:
            hi += 1
        out.append(rec[bounds[lo]:bounds[hi]])
        lo = hi
    return out


class _RecordSource:
    def __init__(self, rank, lengths):
        self.rank, self.lengths = rank, lengths

    def harvest_record_chunks(self, max_plies):
        return iter(_chunks(self.rank, self.lengths, max_plies))


def _record_worker(rank, world, port, rounds, cap, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from chinesechesszero_amd.replay import RecordGatherer, exchange_finished_games
        g = RecordGatherer(cap, "cpu")
        bad, total = [], 0
        assert g.bytes_per_exchange() == -(-(64 + cap * 880) // 64) * 64
        for x, lengths in enumerate(rounds):
            parts = []
            for union, games in exchange_finished_games(_RecordSource(rank, lengths[rank]), g, len(lengths[rank])):
                parts.append((union.clone(), games, g.collectives))
            chunks = [_chunks(r, lengths[r], cap) for r in range(world)]
            want_rounds = max(1, max(len(c) for c in chunks))
            if len(parts) != want_rounds:
                bad.append((x, "rounds", len(parts), want_rounds))
            if sum(p[1] for p in parts) != sum(len(l) for l in lengths):
                bad.append((x, "games"))
            if any(p[2] != 1 for p in parts):           # every iteration of the exchange is exactly ONE collective
                bad.append((x, "collectives", [p[2] for p in parts]))
            # the union: round-major, rank-major inside a round, bytes untouched
            exp = [chunks[r][i] for i in range(want_rounds) for r in range(world) if i < len(chunks[r])]
            exp = torch.cat(exp) if exp else torch.empty((0, 880), dtype=torch.uint8)
            union = torch.cat([p[0] for p in parts])
            if not torch.equal(union, exp):
                bad.append((x, "bytes", tuple(union.shape), tuple(exp.shape)))
            for part, _, _ in parts:                    # every part holds whole games only: it expands on its own
                if part.shape[0]:
                    t = part[:, 96:98].contiguous().view(torch.int16).view(-1)
                    T = part[:, 98:100].contiguous().view(torch.int16).view(-1)
                    if int(t[0]) != 0 or int(t[-1]) != int(T[-1]) - 1:
                        bad.append((x, "cut game"))
            total += int(union.shape[0])
        q.put((rank, not bad, total if not bad else bad))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, False, traceback.format_exc()[-1500:]))


def test_all_gather_of_compact_records_world2():
    """The exchange format of round 3: 880-byte ply records, one collective per exchange while every rank's finished plies fit
    the slot; further rounds are cut BETWEEN games. The gathered bytes are the ranks' records, untouched."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    # per exchange: game lengths per rank. Uneven, an empty rank, nobody, and a rank that needs three rounds (cap 16)
    rounds = [((5, 3), (7,)), ((), (4, 4)), ((), ()), ((9, 6, 2, 11, 5), (16,))]
    procs = [ctx.Process(target=_record_worker, args=(r, world, port, rounds, 16, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert [r[1] for r in res] == [True, True], res
    assert res[0][2] == res[1][2] == sum(sum(l) for rr in rounds for l in rr)  # every rank holds the union


def test_board_id_streams_independent_of_gpu_count():
    """RNG streams are keyed by the GLOBAL board id: rank r's board b uses id r*B+b (oracle twin of the device sampler)."""
    import oracle
    pi = np.full(44, 1 / 44)
    whole = [oracle.det_sample(7, gid, 0, pi)[0] for gid in range(8)]
    shards = [oracle.det_sample(7, r * 4 + b, 0, pi)[0] for r in range(2) for b in range(4)]
    assert whole == shards and len(set(whole)) > 1


def _bcast_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from chinesechesszero_amd.net import PolicyValueNet
        from chinesechesszero_amd.replay import broadcast_model
        torch.manual_seed(10 + rank)  # different weights on every rank before the reload
        pvn = PolicyValueNet(use_gpu=False, device="cpu", num_channels=8, resblocks_num=1)
        for m in pvn.policy_value_net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.fill_(float(rank + 1))
        broadcast_model(pvn, src=0)
        sd = pvn.policy_value_net.state_dict()
        digest = float(sum(v.double().abs().sum() for v in sd.values()))
        q.put((rank, True, digest, float(sd["conv_block_bn.running_mean"][0])))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        q.put((rank, False, repr(e), 0.0))


def test_model_hot_reload_broadcast_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bcast_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] for r in res), res
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3] == 1.0  # rank 1 now holds rank 0's weights and BN stats


class _FakeEngine:
    def __init__(self):
        self.device = torch.device("cpu")
        self.leaf_input = torch.zeros((4, 17, 7, 10, 9), dtype=torch.float16)
        self.over = np.zeros(4, np.uint8)

    def game_status(self):
        return {"over": self.over.copy()}

    def check_healthy(self):
        pass


class _FakeSelfPlay:
    """Stands in for BatchedSelfPlay on the CPU: scripted finished games, chunked rows."""

    def __init__(self, rank, script):
        self.engine = _FakeEngine()
        self.rank = rank
        self.script = script  # per move: list of chunk sizes this rank harvests
        self.move = -1

    def run_move(self, on_playout=None):
        self.move += 1
        self.engine.over[:] = 0
        self.engine.over[:len(self.script[self.move])] = 1 if self.script[self.move] else 0

    def harvest_chunks(self, max_rows):
        for i, n in enumerate(self.script[self.move]):
            s, p, z = _rows(self.rank * 10 + self.move * 3 + i, n)
            yield s, p, z
        self.engine.over[:] = 0


def _collect_worker(rank, world, port, tmp, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from chinesechesszero_amd.collect import CollectPipeline
        from chinesechesszero_amd.replay import TupleGatherer
        cp = CollectPipeline.__new__(CollectPipeline)
        from chinesechesszero_amd.collect import TupleSink
        cp.sink = TupleSink(os.path.join(tmp, f"rank{rank}"))
        cp.iters = 0
        cp.n_boards = 4
        cp.finalize_every = 0
        cp._finalized_at = 0
        cp.on_playout = None
        cp.load_model = lambda: None
        # rank 0: move 0 -> two chunks (5, 2 rows); move 1 -> nothing; rank 1: move 0 -> nothing; move 1 -> three chunks
        script = {0: [[5, 2], []], 1: [[], [3, 1, 4]]}[rank]
        cp.selfplay = _FakeSelfPlay(rank, script)
        g = TupleGatherer(4, "cpu")
        cp.collect_batched(2, gatherer=g)
        collectives = g.collectives  # of the last exchange
        assert not os.path.exists(os.path.join(tmp, f"rank{rank}", "winners.npy"))  # rows stay in shards while collecting
        cp.sink.finalize()
        n = int(np.load(os.path.join(tmp, f"rank{rank}", "winners.npy")).shape[0]) if os.path.exists(os.path.join(tmp, f"rank{rank}", "winners.npy")) else 0
        q.put((rank, True, n, cp.iters))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, False, traceback.format_exc()[-1500:], 0))


def test_collect_batched_gather_loop_world2(tmp_path):
    """Ranks finish different numbers of games in different moves: the exchange loop stays aligned and the union
    of the shards lands in rank 0's store (what N reference collectors would append to one data file)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_collect_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] for r in res), res
    assert res[0][2] == 5 + 2 + 3 + 1 + 4 and res[1][2] == 0   # rank 0 stores the union, rank 1 nothing
    assert res[0][3] == res[1][3] == 2 + 3                     # finished "games" (boards flagged over) counted globally


def test_game_aligned_chunks_cut_between_games_only():
    """engine.game_aligned_chunks (what bounds the dense temporaries when records are expanded): chunks hold whole games, at
    most the asked number of records unless ONE game is longer; nothing is lost or reordered."""
    from chinesechesszero_amd.engine import game_aligned_chunks, rows_of_records
    lengths = (5, 3, 9, 1, 1, 12, 4)
    rec = _records(3, lengths)
    for cap in (1, 4, 5, 8, 9, 13, 35, 100):
        parts = list(game_aligned_chunks(rec, cap))
        assert torch.equal(torch.cat(parts), rec)
        for p in parts:
            t = p[:, 96:98].contiguous().view(torch.int16).view(-1)
            T = p[:, 98:100].contiguous().view(torch.int16).view(-1)
            assert int(t[0]) == 0 and int(t[-1]) == int(T[-1]) - 1                  # starts and ends on a game boundary
            assert p.shape[0] <= cap or p.shape[0] == int(T[0])                     # over the cap only for one long game
    assert rows_of_records(7, 0) == 14 and rows_of_records(7, 2) == 7             # mirror images double the rows; CCZ_FLAG_NO_MIRROR does not
