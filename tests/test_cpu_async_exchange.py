"""CPU: replay.AsyncRecordExchange under gloo -- ranks that never wait for each other (round 5).

The reference's collectors are independent processes (collect.py:181-183, README.md:31-48); these tests pin that a slow rank
delays DATA only: its peers' loops keep their pace, every record still arrives exactly once, in order, on every rank."""
import os
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_cpu_distributed import _free_port, _records


def _games(rank, move, lengths):
    """Records of the games a rank finishes in one move; byte 0 of every record carries (rank, move) so streams can be told apart."""
    rec = _records(1000 * rank + move, lengths)
    rec[:, 0] = rank
    rec[:, 1] = move
    return rec


def _split(union, rows_per_rank):
    out, lo = [], 0
    for n in rows_per_rank:
        out.append(union[lo:lo + n])
        lo += n
    return out


def _run_rank(rank, world, port, script, cap, q, sleep_at=None, steps_per_move=10, step_s=0.01, timeout_s=60.0):
    """script[rank] = per move: tuple of game lengths. Between two moves a rank does ``steps_per_move`` ticks of ``step_s``."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from chinesechesszero_amd.replay import AsyncRecordExchange
        ex = AsyncRecordExchange(cap, "cpu", timeout_s=timeout_s)
        streams = [[] for _ in range(world)]   # what arrived from every source rank, in arrival order
        games = 0
        calls = []

        def take(done):
            nonlocal games
            for x in done:
                games += x.games
                for k, seg in enumerate(_split(x.union, x.rows_per_rank)):
                    if seg.shape[0]:
                        streams[k].append(seg.clone())

        t_start = time.perf_counter()
        for move, lengths in enumerate(script[rank]):
            if sleep_at is not None and sleep_at == (rank, move):
                time.sleep(2.0)                       # a rank that falls behind (a trainer sharing its GPU, a long move ...)
            for _ in range(steps_per_move):
                time.sleep(step_s)
                t0 = time.perf_counter()
                take(ex.tick())
                calls.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            take(ex.post([_games(rank, move, lengths)] if lengths else [], games=len(lengths)))
            calls.append(time.perf_counter() - t0)
        loop_s = time.perf_counter() - t_start
        take(ex.flush())
        want = [torch.cat([_games(r, m, l) for m, l in enumerate(script[r]) if l] or [torch.empty((0, 880), dtype=torch.uint8)]) for r in range(world)]
        got = [torch.cat(s) if s else torch.empty((0, 880), dtype=torch.uint8) for s in streams]
        same = all(torch.equal(a, b) for a, b in zip(want, got))
        q.put((rank, same, {"loop_s": loop_s, "max_call_s": max(calls), "games": games, "issued": ex.issued, "completed": ex.completed,
                            "max_backlog": ex.max_backlog_plies, "host_s": ex.host_seconds}))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, False, traceback.format_exc()[-2000:]))


def _spawn(world, target, args_of_rank, timeout=180):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args_of_rank(r) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=timeout) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
    return res


def _launch(world, script, cap, **kw):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_rank, args=(r, world, port, script, cap, q), kwargs=kw) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
    return res


def test_a_slow_rank_does_not_stall_its_peer_world2():
    """Rank 1 sleeps 2 s in front of its third move. Rank 0 finishes all eight of its moves in well under that, no call into the
    exchange ever takes long, and after the drain both ranks hold both ranks' records, complete and in order."""
    script = {0: [(5, 3), (7,), (), (4, 4), (2,), (6, 1), (3,), (9,)],
              1: [(2,), (8,), (3, 3), (), (5,), (1, 1, 1), (4,), (2, 2)]}
    res = _launch(2, script, 64, sleep_at=(1, 2))
    assert all(r[1] is True for r in res), res
    r0, r1 = res[0][2], res[1][2]
    assert r0["loop_s"] < 1.6 < r1["loop_s"], (r0, r1)        # rank 0: 8 moves x 10 steps x 10 ms; rank 1 additionally slept 2 s
    assert r0["max_call_s"] < 0.5, r0                          # nothing rank 0 called waited for rank 1
    assert r0["games"] == r1["games"] == sum(len(l) for s in script.values() for l in s)
    assert r0["issued"] == r1["issued"] == r0["completed"]     # the ranks paired the same exchanges
    assert r0["max_backlog"] > 8                               # rank 0's games queued up while rank 1 slept (data waited, the loop did not)


def test_eight_ranks_uneven_loads_empty_rank_small_slot():
    """World 8: different numbers of moves per rank (paces differ), one rank that never finishes a game, loads that overflow the
    slot (whole games carry over to the next exchange). Every rank ends with every rank's records, complete and in order."""
    rs = np.random.RandomState(5)
    script = {}
    for r in range(8):
        moves = 3 + (r % 4)
        script[r] = [tuple(int(x) for x in rs.randint(1, 9, size=rs.randint(0, 4))) for _ in range(moves)]
    script[3] = [(), (), ()]                                   # the empty rank
    script[6] = [(12, 12, 12, 12, 12), (12,)]                  # 72 plies against a 32-ply slot: three exchanges' worth
    res = _launch(8, script, 32, steps_per_move=4)
    assert all(r[1] is True for r in res), [r for r in res if r[1] is not True]
    total = sum(len(l) for s in script.values() for l in s)
    assert all(r[2]["games"] == total for r in res)
    assert len({r[2]["issued"] for r in res}) == 1 and all(r[2]["issued"] == r[2]["completed"] for r in res)


def _abort_rank(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from chinesechesszero_amd.replay import AsyncRecordExchange, GatherAborted
        ex = AsyncRecordExchange(16, "cpu", timeout_s=60)
        ex.post([_games(rank, 0, (3,))], games=1)
        ex.flush()                                             # exchange 0, 1: fine
        raised_at = None
        try:
            ex.post([_games(rank, 1, (40,) if rank == 5 else (2,))], games=1)   # rank 5: one game longer than the slot
            ex.flush()
        except GatherAborted as e:
            raised_at = (ex.completed, str(e))
        q.put((rank, raised_at is not None, raised_at))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, False, traceback.format_exc()[-2000:]))


def test_abort_flag_raises_on_all_eight_ranks_after_the_same_exchange():
    res = _spawn(8, _abort_rank, lambda r: ())
    assert all(r[1] is True for r in res), res
    assert len({r[2][0] for r in res}) == 1                    # the same exchange index everywhere
    assert all("[5]" in r[2][1] for r in res)
    assert "longer than the exchange slot" in res[5][2][1]     # the rank that aborted says why


def _timeout_rank(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from chinesechesszero_amd.replay import AsyncRecordExchange
        ex = AsyncRecordExchange(16, "cpu", timeout_s=1.0)
        msg = None
        if rank == 0:
            ex.post([_games(0, 0, (3,))], games=1)
            t0 = time.perf_counter()
            try:
                while time.perf_counter() - t0 < 10:
                    ex.tick()
                    time.sleep(0.01)
            except RuntimeError as e:
                msg = (time.perf_counter() - t0, str(e))
        else:
            time.sleep(5.0)                                    # never announces (long enough for a loaded machine to reach the timeout first)
        q.put((rank, True, msg))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, False, traceback.format_exc()[-2000:]))


def test_a_rank_that_never_announces_is_named_after_the_timeout():
    res = _spawn(2, _timeout_rank, lambda r: ())
    assert all(r[1] is True for r in res), res
    waited, msg = res[0][2]
    assert 1.0 <= waited < 5.0 and "rank(s) [1]" in msg and "did not announce" in msg


def test_group_of_one_hands_the_backlog_straight_over():
    from chinesechesszero_amd.replay import AsyncRecordExchange
    ex = AsyncRecordExchange(16, "cpu")
    a, b = _games(0, 0, (3, 2)), _games(0, 1, (4,))
    done = ex.post([a], games=2)
    assert len(done) == 1 and torch.equal(done[0].union, a) and done[0].games == 2
    assert ex.tick() == []
    done = ex.post([b], games=1) + ex.flush()
    assert len(done) == 1 and torch.equal(done[0].union, b)


def test_tick_until_polls_the_event_and_hands_over_what_completes():
    from chinesechesszero_amd.replay import AsyncRecordExchange

    class Ev:
        def __init__(self, n):
            self.n = n

        def query(self):
            self.n -= 1
            return self.n < 0

    ex = AsyncRecordExchange(16, "cpu")
    ev = Ev(3)
    assert ex.tick_until(ev, poll_s=0.001) == [] and ev.n < 0         # nothing pending: polled until the event was done
    ex._backlog.append(_games(0, 0, (3,)))                            # (a record that arrives while the rank waits for its GPU)
    ex._backlog_plies = 3
    got = ex.tick_until(Ev(50), poll_s=0.001)
    assert len(got) == 1 and got[0].union.shape[0] == 3               # handed over at once, before the event completed


def test_board_partition_prefix_sums():
    """Per-rank board counts (a lighter rank 0 next to the trainer): global board ids are a prefix sum, so every board keeps its
    RNG stream whatever the split."""
    from chinesechesszero_amd.launch import board_partition
    assert board_partition(4, 4096) == ([4096] * 4, [0, 4096, 8192, 12288])
    assert board_partition(4, 4096, 512) == ([512, 4096, 4096, 4096], [0, 512, 4608, 8704])
    assert board_partition(1, 64, 16) == ([16], [0])
    with pytest.raises(ValueError):
        board_partition(2, 64, 0)


def test_the_collector_keeps_the_exchange_moving_during_a_move_and_does_not_lose_its_errors():
    """CollectPipeline._run_move_ticking: tick() from the search's on_playout callback (the search swallows what a callback raises,
    mcts.py:156-159 -- an exchange error must still reach the caller), the user's own callback still served."""
    from chinesechesszero_amd.collect import CollectPipeline

    class Engine:
        device = torch.device("cpu")

    class FakeSelfPlay:
        n_playout, _sim, engine = 8, 0, Engine()

        def advance(self, steps, on_playout=None, boundary=None):
            for k in range(steps):
                try:
                    on_playout(1)
                except Exception:
                    pass                       # what selfplay.advance does with a callback's exception
            return boundary()

        def finish_move(self):
            return "moves"

    class FakeExchange:
        def __init__(self, fail_at=None):
            self.ticks, self.fail_at = 0, fail_at

        def tick(self):
            self.ticks += 1
            if self.fail_at == self.ticks:
                raise RuntimeError("exchange 3: rank(s) [1] did not announce within 180 s")
            return []

        def tick_until(self, ev):
            return []

    cp = CollectPipeline.__new__(CollectPipeline)
    cp.selfplay, seen = FakeSelfPlay(), []
    cp.on_playout = lambda k: seen.append(k)
    ex = FakeExchange()
    assert cp._run_move_ticking(ex) == "moves" and ex.ticks == 8 and len(seen) == 8
    ex = FakeExchange(fail_at=3)
    with pytest.raises(RuntimeError, match="did not announce"):
        cp._run_move_ticking(ex)
    assert ex.ticks == 3                                                  # after the failure the exchange is left alone


def test_the_collector_ends_a_run_in_the_right_order():
    """CollectPipeline.run: a normal end drains the exchange, then merges -- or leaves the merge to the caller
    (finalize=False: the multi-rank CLI merges after its process group is gone, so no peer waits in a barrier for rank 0's minutes
    of expansion); an error inside a multi-rank run goes straight up -- no blocking drain on peers that will never announce, no
    second error over the first, no merge; a single process still merges what it has; finalize_every is ignored while an
    exchange with other ranks is live."""
    from chinesechesszero_amd.collect import CollectPipeline

    class Sink:
        def __init__(self):
            self.calls, self.games = [], 0

        def finalize(self):
            self.calls.append("finalize")

    class Exchange:
        def __init__(self, world):
            self.world, self.calls = world, []

        def flush_iter(self):
            self.calls.append("drain")
            return iter(())

    def pipe(world, fail_at=None, finalize_every=0):
        cp = CollectPipeline.__new__(CollectPipeline)
        cp.sink, cp.gatherer, cp.selfplay, cp.n_boards, cp.episode_len = Sink(), (Exchange(world) if world else None), object(), 64, 0
        cp.finalize_every, cp._finalized_at, cp.iters, n = finalize_every, 0, 0, [0]

        def collect_data(is_shown=False):
            n[0] += 1
            if fail_at == n[0]:
                raise RuntimeError("exchange 3: rank(s) [1] did not announce within 180 s")
            cp.sink.games += 10
            cp._maybe_finalize()
            return n[0]
        cp.collect_data = collect_data
        return cp

    cp = pipe(world=2)
    cp.run(max_calls=3)
    assert cp.gatherer.calls == ["drain"] and cp.sink.calls == ["finalize"]
    cp = pipe(world=2)
    cp.run(max_calls=3, finalize=False)                    # the CLI: barrier + destroy_process_group come next, THEN sink.finalize()
    assert cp.gatherer.calls == ["drain"] and cp.sink.calls == []
    cp = pipe(world=2, fail_at=2)
    with pytest.raises(RuntimeError, match="did not announce"):
        cp.run(max_calls=3)
    assert cp.gatherer.calls == [] and cp.sink.calls == []  # straight up: launch.guarded ends the process
    cp = pipe(world=0, fail_at=2)
    with pytest.raises(RuntimeError):
        cp.run(max_calls=3)
    assert cp.sink.calls == ["finalize"]                   # one process: what was collected is merged before the error goes up
    cp = pipe(world=2, finalize_every=10)
    cp.run(max_calls=3, finalize=False)
    assert cp.sink.calls == []                             # no merge on the launch thread while peers expect announcements
    cp = pipe(world=0, finalize_every=10)
    cp.run(max_calls=3)
    assert cp.sink.calls == ["finalize"] * 4               # one process: every 10 games, and at the end
