"""Multi-GPU exchange: all-gather of finished games.

Games shard embarrassingly over ranks (one process per GPU, own engine, RNG streams keyed by the
global board id); the ONLY collective is this all-gather, backend ``nccl`` (= RCCL over xGMI) on
GPUs and ``gloo`` in the CPU tests. One fused, fixed-shape buffer per rank (header with the counts + the
payload), so that every rank issues the same ONE collective per exchange regardless of how many games
finished where. The payload is the COMPACT game record (880 B per ply: position, sparse pi, winner), not
the dense (state, pi, z) rows it stands for (2 x 29,768 B per ply): the rows are rebuilt by
``ccz_expand_records`` on the receiving side.
The reference has no counterpart: its "exchange" is N collector processes appending to one HDF5
file (collect.py:146-167); parity = the union of shards equals what N collectors would append.
"""
from __future__ import annotations

import time
from typing import NamedTuple

import torch
import torch.distributed as dist

STATE_ELEMS = 17 * 7 * 10 * 9
NMOVES = 2086
REC_BYTES, REC_HDR = 880, 96  # compact ply record, include/cczero.h CCZ_REC_BYTES / CCZ_REC_HDR


class GatherAborted(RuntimeError):
    """Raised on EVERY rank of an exchange in which some rank could not take part (e.g. one game longer than the slot)."""


class _FusedGather:
    """ONE ``all_gather_into_tensor`` per round: every rank sends one fixed-size uint8 slot ``[header 64 B | payload]``.

    header = int64 x 8: items in this round, items this rank still holds after it, ``more`` flag of the caller, ``user``
    counter (finished games), ``abort`` (this rank cannot take part: every rank raises after the collective, together), 3 spare. Counts therefore travel inside the same collective as the payload; after it ONE
    small device-to-host copy (world x 64 B) tells every rank what each slot holds. The shape is fixed (padded) so that
    every rank issues the same collective whatever finished where."""

    HEADER = 64

    def __init__(self, payload_bytes: int, device, group=None, always_collective: bool = False):
        """``always_collective``: issue the collective even in a group of one (exercises the RCCL path on a single GPU)."""
        self.always_collective = bool(always_collective)
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.slot_bytes = -(-(self.HEADER + int(payload_bytes)) // 64) * 64
        d = self.device
        self._send = torch.zeros((self.slot_bytes,), dtype=torch.uint8, device=d)
        self._recv = torch.zeros((self.world * self.slot_bytes,), dtype=torch.uint8, device=d)
        self._hdr_host = torch.zeros((8,), dtype=torch.int64)
        if d.type == "cuda":
            self._hdr_host = self._hdr_host.pin_memory()
        # statistics of the last gather() (bench.py reports them)
        self.any_more = False
        self.user_sum = 0
        self.rounds = 0
        self.collectives = 0
        self.seconds = 0.0
        self.rows_per_rank: list[int] = []
        self._abort_why = ""

    def bytes_per_exchange(self) -> int:
        """Bytes one rank sends in one round (it receives (world - 1) x this)."""
        return self.slot_bytes

    def _solo(self) -> bool:
        return self.world == 1 and not (self.always_collective and dist.is_initialized())

    def _round(self, m: int, left: int, more: bool, user: int, abort: int = 0):
        """Header in, the collective, headers out (the one host sync): int64 [world, 8] on the host. A rank that cannot go on
        (``abort`` != 0) still takes part in THIS collective, so that nobody is left waiting in it; afterwards every rank raises."""
        self._hdr_host[0] = m
        self._hdr_host[1] = left
        self._hdr_host[2] = 1 if more else 0
        self._hdr_host[3] = int(user)
        self._hdr_host[4] = int(abort)
        self._send[:self.HEADER].view(torch.int64).copy_(self._hdr_host, non_blocking=True)
        dist.all_gather_into_tensor(self._recv, self._send, group=self.group)
        self.collectives += 1
        self.rounds += 1
        heads = self._recv.view(self.world, self.slot_bytes)[:, :self.HEADER].contiguous().view(torch.int64).view(self.world, 8).cpu()
        for k in range(self.world):
            self.any_more |= bool(int(heads[k, 2]))
            self.user_sum += int(heads[k, 3])
        bad = [k for k in range(self.world) if int(heads[k, 4])]
        if bad:
            raise GatherAborted(f"rank(s) {bad} aborted the exchange" + (f": {self._abort_why}" if self.rank in bad and self._abort_why else ""))
        return heads


class RecordGatherer(_FusedGather):
    """All-gather of finished games as COMPACT PLY RECORDS (``include/cczero.h`` CCZ_REC_*, 880 B per ply): one
    collective per exchange at the benchmark workload.

    A move of 4096 boards ends ~100 games = ~14 k plies = 12.5 MB of records per rank, against 847 MB for the same games
    as 28 k dense rows of 29,768 B (fp16 one-hot planes + float32 pi[2086]); the default slot holds 32,768 plies
    (28.8 MB). The dense rows are rebuilt on the receiving side by ``ccz_expand_records`` (:func:`engine.expand_records`,
    :meth:`ReplayBuffer.append_records`), byte for byte what ``ccz_harvest`` writes. Games are never cut: a round carries
    whole games only, so every received segment expands on its own."""

    def __init__(self, capacity_plies: int = 32768, device="cpu", group=None, always_collective: bool = False):
        self.cap = int(capacity_plies)
        if self.cap <= 0:
            raise ValueError("capacity_plies must be positive")
        super().__init__(self.cap * REC_BYTES, device, group, always_collective)

    def _payload(self, buf, k):
        base = k * self.slot_bytes + self.HEADER
        return buf[base: base + self.cap * REC_BYTES].view(self.cap, REC_BYTES)

    @staticmethod
    def _whole_games(records: torch.Tensor, lo: int, room: int) -> int:
        """How many of records[lo:] fit ``room`` plies without cutting a game (reads ONE record's header, and only when
        the segment does not fit as a whole)."""
        n = int(records.shape[0]) - lo
        if n <= room:
            return n
        # records[lo] starts a game, so the game of records[lo + room] (the first record that does not fit) starts at
        # lo + room - t: cut in front of it
        t = int(records[lo + room, REC_HDR:REC_HDR + 2].cpu().view(torch.int16).item()) & 0xffff
        if t >= room:
            raise ValueError(f"one game is longer than the exchange slot ({room} plies): raise capacity_plies")
        return room - t

    def gather(self, records: torch.Tensor, more: bool = False, user: int = 0) -> torch.Tensor:
        """Every rank passes its new records uint8 [P, 880] (possibly none); every rank gets all of them, rank-major,
        as one uint8 [sum P, 880] tensor on this gatherer's device. ``more`` / ``user``: as :meth:`TupleGatherer.gather`.
        The result may be a VIEW of the receive buffer (no copy when one rank's segment is all there is): consume it -- or
        enqueue its consumer on the current stream -- before the next ``gather``."""
        n = int(records.shape[0])
        self.rounds = self.collectives = 0
        self.seconds = 0.0
        if self._solo():
            self.any_more, self.user_sum, self.rows_per_rank = bool(more), int(user), [n]
            return records
        t0 = time.perf_counter()
        outs = [[] for _ in range(self.world)]
        totals = [0] * self.world
        self.any_more, self.user_sum = False, 0
        mine = self._payload(self._send, 0)
        lo = 0
        while True:
            abort = 0
            try:
                m = self._whole_games(records, lo, self.cap) if n > lo else 0
            except ValueError as e:   # this rank holds a game the slot cannot carry: the peers are (or will be) in the collective --
                m, abort, self._abort_why = 0, 1, str(e)   # join it with the abort flag up, so that all ranks raise together
            if m:
                mine[:m].copy_(records[lo:lo + m])
            heads = self._round(m, n - lo - m, more, user if self.rounds == 0 else 0, abort)
            again = bool((heads[:, 1] > 0).any())  # some rank still holds records: another round of the same shape follows
            for k in range(self.world):
                mk = int(heads[k, 0])
                if mk:
                    seg = self._payload(self._recv, k)[:mk]
                    outs[k].append(seg.clone() if again else seg)  # the next round overwrites the receive buffer
                    totals[k] += mk
            lo += m
            if not again:
                break
        self.rows_per_rank = totals
        flat = [t for k in range(self.world) for t in outs[k]]
        out = records[:0] if not flat else (flat[0] if len(flat) == 1 else torch.cat(flat))
        self.seconds = time.perf_counter() - t0  # staging copy + collective(s) + the header read that waits for them
        return out


def exchange_finished_games(source, gatherer: RecordGatherer, done: int):
    """One exchange step of a move boundary: harvest this rank's finished games as compact records and all-gather them.

    Every rank calls this once per move (with ``done`` = its own count of finished games, possibly 0) and iterates it to
    the end: each iteration is ONE collective and yields ``(records uint8 [P, 880] on gatherer.device -- the union over
    ranks, rank-major, whole games --, finished games summed over ranks)``. There is more than one iteration only when
    some rank finished more plies than the gatherer's slot holds; the "more" flag and the game count ride in the header
    of the same collective (no extra all-reduce). ``source``: anything with ``harvest_record_chunks(max_plies)``."""
    longest = getattr(getattr(source, "engine", source), "max_plies", None)
    if longest is not None and int(longest) > gatherer.cap:
        # the same test on every rank (they share the configuration), BEFORE any collective: nobody is left waiting in one
        raise ValueError(f"exchange slot of {gatherer.cap} plies is smaller than the longest game the engine records ({longest} plies): "
                         f"RecordGatherer(capacity_plies >= max_plies)")
    it = iter(source.harvest_record_chunks(gatherer.cap)) if done else iter(())
    chunk = next(it, None)
    first = True
    empty = torch.empty((0, REC_BYTES), dtype=torch.uint8, device=gatherer.device)
    while True:
        nxt = next(it, None) if chunk is not None else None
        mine = empty if chunk is None else chunk.to(gatherer.device)
        union = gatherer.gather(mine, more=nxt is not None, user=done if first else 0)
        first = False
        yield union, gatherer.user_sum
        if not gatherer.any_more:
            return
        chunk = nxt


class Exchanged(NamedTuple):
    """One completed exchange of :class:`AsyncRecordExchange`."""
    union: torch.Tensor          # uint8 [P, 880]: every rank's records, rank-major, whole games
    games: int                   # finished games announced with them, summed over ranks
    rows_per_rank: list          # records per source rank (the union's segments)
    index: int                   # exchange number (the same on every rank)


class AsyncRecordExchange(_FusedGather):
    """The all-gather of finished games with NO rank ever waiting for another one (round 5).

    The reference's collectors and its trainer are separate processes that never wait on each other (README.md:31-48,
    collect.py:181-183: N shell commands appending to one file). :class:`RecordGatherer` met every peer once per move inside a
    blocking collective + header read, so a job ran at the pace of its slowest rank (the one that shares its GPU with the trainer).
    Here a rank only ever does three non-blocking things:

    * :meth:`post` -- at its move boundary: the finished games (compact ply records, whole games) join a local BACKLOG in
      device memory; as soon as no exchange of its own is in flight the rank STAGES what fits the slot into its send buffer
      and ANNOUNCES exchange ``j`` on the job's key-value store (the rendezvous TCPStore): how many plies it will send, how many
      it holds back, the finished-game counter, and the closing / abort flags;
    * :meth:`tick` -- any time (bench.py: after every simulation step): when ALL ranks have announced ``j`` every rank reads the
      same announcements and therefore takes the same decision: nobody has anything -> the exchange is VIRTUAL (no collective at
      all); somebody aborted -> every rank raises; otherwise ONE ``all_gather_into_tensor(async_op=True)`` of the smallest
      power-of-two slot that holds the largest announcement is issued from a side stream (it depends on the harvest only, not on
      the simulation steps queued on the main stream). A completed exchange is handed over as an :class:`Exchanged`
      ``(union = records uint8 [P, 880], rank-major, whole games; games; rows_per_rank; index)``;
    * the collective itself runs on the process group's own stream while the search goes on.

    Exchanges pair by their index ``j``, not by move number: a fast rank posts several moves into one exchange, a slow rank
    delays DATA, never its peers' launch loops. The store is the control plane (counts, flags), the collective pure payload.
    The handshake also keeps the collective's kernel from spinning on a fast rank's CUs while a slow rank is still a move away:
    it is launched only once every rank is known to be within one ``tick`` of launching it too. At most ONE exchange is in
    flight; the receive slot is double-buffered, so a handed-over union stays valid until the second-next exchange is issued
    (consume it, or enqueue its consumer on the current stream, before the next call into the exchange).

    What stays of :class:`RecordGatherer`: the fused ``[header 64 B | payload]`` slot, whole games only, and the abort rule --
    a rank that cannot take part (one game longer than the slot) says so in its announcement and EVERY rank raises
    :class:`GatherAborted` at that exchange. A peer that never announces makes :meth:`tick` raise after ``timeout_s`` (naming the
    missing ranks) instead of leaving anyone inside a collective. :meth:`flush_iter` (blocking; end of a run, or before any other
    collective of the job) drains: exchanges are repeated until every rank announces "closing, nothing to send, nothing held
    back" -- all ranks read the same announcements, so all stop at the same exchange, and that last one costs no collective.
    """

    READY_POLL_S = 0.02   # a waiting rank asks the store at most this often (one ~0.1 ms round trip)
    MIN_CLASS = 64        # smallest payload class (plies) of a collective
    _instances = 0

    def __init__(self, capacity_plies: int = 32768, device="cpu", group=None, always_collective: bool = False,
                 store=None, timeout_s: float = 180.0, name: str = "ccz_xchg"):
        self.cap = int(capacity_plies)
        if self.cap <= 0:
            raise ValueError("capacity_plies must be positive")
        super().__init__(self.cap * REC_BYTES, device, group, always_collective)
        self._recv2 = [self._recv, torch.zeros_like(self._recv)]
        self.timeout_s = float(timeout_s)
        self._name = name
        self._store = store
        if self._store is None and not self._solo() and dist.is_initialized():
            from torch.distributed.distributed_c10d import _get_default_store
            # (every rank builds its exchanges in the same order: the instance counter keeps their keys apart)
            AsyncRecordExchange._instances += 1
            self._store = dist.PrefixStore(f"{name}{AsyncRecordExchange._instances}", _get_default_store())
        self._xs = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._backlog: list[torch.Tensor] = []
        self._backlog_plies = 0
        self._games_pending = 0
        self._want = False           # this rank has something to say (a boundary was posted / it is draining) and has not announced it yet
        self._staged = None          # (plies staged in the send slot, abort flag, games) of the announced exchange
        self._harvest_ev = None
        self._work = None
        self._inflight = None        # (index, receive buffer, slot bytes, announcements) of the exchange in flight
        self._announced = -1         # highest exchange index this rank has announced
        self._announced_at = 0.0
        self._last_ready_check = 0.0
        self._closing = False
        self._all_closed = False
        self._handed_over = None     # receive slot whose union was handed to the caller by the last call (its consumer is being enqueued)
        self._consumed_ev = [None, None]   # per receive slot: recorded on the main stream one call after the hand-over
        self.issued = 0              # exchanges decided so far, virtual ones included (the next one has index ``issued``)
        self.completed = 0
        self.virtual = 0             # of those: exchanges in which no rank had anything to send (no collective)
        self.moves_posted = 0
        # statistics (bench.py reports them): host seconds spent inside post()/tick() -- what this rank's launch loop lost
        self.host_seconds = 0.0
        self.max_call_s = 0.0
        self.max_backlog_plies = 0
        self.plies_sent = 0
        self.bytes_sent = 0
        self.last_slot_bytes = 0
        self.last_plan = None

    # ---- the non-blocking calls ------------------------------------------------------------------------------------------
    def post(self, chunks, games: int = 0):
        """This rank's finished games of one move boundary: ``chunks`` = iterable of uint8 [P, 880] tensors, whole games each
        (``engine.harvest_record_chunks``), possibly empty. Never waits for a peer. Returns what :meth:`tick` returns."""
        t0 = time.perf_counter()
        if isinstance(chunks, torch.Tensor):
            chunks = [chunks]
        for c in chunks:
            if int(c.shape[0]) == 0:
                continue
            c = c if c.device == self.device else c.to(self.device)
            self._backlog.append(c)
            self._backlog_plies += int(c.shape[0])
        if self._xs is not None:   # the side stream may read the new records only after the harvest that wrote them
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._harvest_ev = ev
        self._games_pending += int(games)
        self.moves_posted += 1
        self.max_backlog_plies = max(self.max_backlog_plies, self._backlog_plies)
        self._want = True
        self._account(t0)
        return self.tick()

    def tick(self, block: bool = False):
        """Advance the exchange without waiting for anybody: complete the collective in flight if it has finished; announce what
        this rank has posted; decide the next exchange if every rank has announced it. Returns the list of completed exchanges
        (:class:`Exchanged`; empty nearly always). ``block``: wait for the collective in flight / for the announcements (used
        by :meth:`flush_iter`)."""
        t0 = time.perf_counter()
        out = []
        self._mark_consumed()
        try:
            if self._solo():
                if self._backlog or self._games_pending:
                    union = self._backlog[0] if len(self._backlog) == 1 else (torch.cat(self._backlog) if self._backlog else
                                                                              torch.empty((0, REC_BYTES), dtype=torch.uint8, device=self.device))
                    out.append(Exchanged(union, self._games_pending, [int(union.shape[0])], self.issued))
                    self.rows_per_rank = [int(union.shape[0])]
                    self.plies_sent += int(union.shape[0])
                    self._backlog, self._backlog_plies, self._games_pending = [], 0, 0
                    self.issued += 1
                    self.completed += 1
                self._all_closed = self._closing
                self._want = False
                return out
            if self._work is not None:
                if block or self._work.is_completed():
                    out.append(self._complete())
                return out
            if self._want and self._announced < self.issued:
                self._announce()
            if self._announced >= self.issued:
                plan = self._everyone_announced(block)
                if plan is not None:
                    done = self._decide(plan)
                    if done is not None:
                        out.append(done)
            return out
        finally:
            self._account(t0)

    def tick_until(self, event, poll_s: float = 0.002):
        """The rank is about to block on the GPU (the launch loop runs ahead of the device; a move boundary starts with a read-back
        that waits for everything queued): wait for ``event`` (a recorded ``torch.cuda.Event``) HERE instead, ticking meanwhile, so
        that an exchange every other rank has announced is not left waiting -- its kernel spinning on the peers' CUs -- until this
        rank's host comes back from its sync. Returns the completed exchanges (consume them before the next call)."""
        out = []
        while not event.query():
            out += self.tick()
            if out:          # hand over at once: the caller consumes before the next tick (receive-slot reuse is ordered on that)
                return out
            time.sleep(poll_s)
        return out

    def flush(self):
        """Blocking drain (end of a run; before any OTHER collective of the job is issued): waits for the exchange in flight and
        repeats exchanges until every rank is closing with nothing left. Returns the completed exchanges, in order; their
        unions are COPIES (the drain may run through more exchanges than the receive slots hold)."""
        return [x._replace(union=x.union.clone()) for x in self.flush_iter()]

    def flush_iter(self):
        """:meth:`flush` as a generator: every completed exchange is yielded before the next one is decided, so the union (a view
        of a receive slot) can be consumed in place. Iterate it to the end."""
        self._closing = True
        self._all_closed = False
        try:
            while True:
                self._want = True
                for x in self.tick(block=True):       # completes the exchange in flight, else announces, waits for all ranks, decides
                    yield x
                if self._all_closed:
                    return
        finally:
            self._closing = False

    def bytes_per_exchange(self) -> int:
        """Bytes this rank sent in its last collective (the slot class of that exchange; the full slot before the first one)."""
        return self.last_slot_bytes or self.slot_bytes

    # ---- internals -------------------------------------------------------------------------------------------------------
    def _mark_consumed(self):
        """The union handed over by the previous call has been consumed -- or its consumer enqueued on the current stream -- by now
        (the caller does that before it calls again): an event on the current stream marks the point after which that receive slot
        may be overwritten. The collective that reuses the slot (two exchanges later) waits for it: by stream order, not by luck."""
        if self._handed_over is not None and self._xs is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._consumed_ev[self._handed_over] = ev
        self._handed_over = None

    def _account(self, t0):
        dt = time.perf_counter() - t0
        self.host_seconds += dt
        self.max_call_s = max(self.max_call_s, dt)

    def _key(self, j, rank=None):
        return f"x{j}" if rank is None else f"x{j}r{rank}"

    def _announce(self):
        """Stage what fits the slot (whole games) into the send buffer and tell the job: ``plies, held back, closing, abort, games``.
        Only while no exchange of this rank is in flight (the send buffer is free). What is posted later waits for the next exchange."""
        j = self.issued
        with (torch.cuda.stream(self._xs) if self._xs is not None else _NullCtx()):
            if self._xs is not None and self._harvest_ev is not None:
                self._xs.wait_event(self._harvest_ev)
            m, abort = self._fill(self._payload_of(self._send))
        self._staged = (m, abort, self._games_pending)
        self._games_pending = 0
        self._want = False
        self._announced = j
        self._announced_at = time.perf_counter()
        if self._store is not None:
            self._store.set(self._key(j, self.rank), f"{m},{self._backlog_plies},{1 if self._closing else 0},{abort},{self._staged[2]}")
            self._store.add(self._key(j), 1)

    def _everyone_announced(self, block: bool):
        """All ranks' announcements of exchange ``issued`` -- int64 [world, 5] (plies, held back, closing, abort, games) -- or None
        while some are missing. One store round trip per poll, rate-limited; ``block`` polls until they are there (or the timeout
        names the ranks that are not)."""
        j = self.issued
        if self._store is None:
            m, abort, games = self._staged
            return [[m, self._backlog_plies, 1 if self._closing else 0, abort, games]]
        while True:
            now = time.perf_counter()
            if block or now - self._last_ready_check >= self.READY_POLL_S:
                self._last_ready_check = now
                if int(self._store.add(self._key(j), 0)) >= self.world:
                    keys = [self._key(j, k) for k in range(self.world)]
                    try:
                        vals = self._store.multi_get(keys)
                    except Exception:
                        vals = [self._store.get(k) for k in keys]
                    return [[int(x) for x in (v.decode() if isinstance(v, (bytes, bytearray)) else str(v)).split(",")] for v in vals]
            if now - self._announced_at > self.timeout_s:
                missing = []
                for k in range(self.world):
                    try:
                        if not self._store.check([self._key(j, k)]):
                            missing.append(k)
                    except Exception:
                        missing.append(k)
                raise RuntimeError(f"exchange {j}: rank(s) {missing} did not announce within {self.timeout_s:.0f} s "
                                   f"(this rank {self.rank} has {self._backlog_plies + (self._staged[0] if self._staged else 0)} plies waiting)")
            if not block:
                return None
            time.sleep(0.0005)     # (the drain sits inside bench.py's timed region: poll the store briskly, one ~0.1 ms round trip each)

    def _decide(self, plan):
        """Every rank holds the same ``plan`` (the announcements of exchange ``issued``) and takes the same branch."""
        j = self.issued
        self.last_plan = plan
        if self._store is not None and self.rank == 0 and j >= 2:   # the announcements of exchange j - 2 are history
            try:
                for key in [self._key(j - 2)] + [self._key(j - 2, k) for k in range(self.world)]:
                    self._store.delete_key(key)
            except Exception:
                pass
        bad = [k for k in range(self.world) if plan[k][3]]
        if bad:
            self.issued += 1
            self.completed += 1
            self._staged = None
            raise GatherAborted(f"rank(s) {bad} aborted the exchange" + (f": {self._abort_why}" if self.rank in bad and self._abort_why else ""))
        most = max(p[0] for p in plan)
        if most == 0:
            # nobody sends anything: no collective. (Games announced without records cannot happen -- they ride with their records --
            # but the counter is passed on all the same.)
            self.issued += 1
            self.completed += 1
            self.virtual += 1
            self._staged = None
            self.rows_per_rank = [0] * self.world
            self.user_sum = sum(p[4] for p in plan)
            self._all_closed = all(p[2] for p in plan) and all(p[1] == 0 for p in plan)
            if self.user_sum:
                return Exchanged(torch.empty((0, REC_BYTES), dtype=torch.uint8, device=self.device), self.user_sum, list(self.rows_per_rank), j)
            return None
        cls = self.MIN_CLASS
        while cls < most:
            cls *= 2
        cls = min(cls, self.cap)
        self._issue(j, plan, -(-(self.HEADER + cls * REC_BYTES) // 64) * 64)
        return None

    def _fill(self, payload) -> tuple[int, int]:
        """Move whole games from the backlog into the send slot: (records placed, abort flag)."""
        m = 0
        while self._backlog and m < self.cap:
            c = self._backlog[0]
            n = int(c.shape[0])
            take = n
            if n > self.cap - m:
                if m:          # a chunk that does not fit behind what is already placed waits for the next exchange
                    break
                try:
                    take = self._whole_games(c, 0, self.cap)
                except ValueError as e:   # one game longer than the slot: say so in the announcement, every rank raises
                    self._abort_why = str(e)
                    return m, 1
            payload[m:m + take].copy_(c[:take], non_blocking=True)
            if self._xs is not None:
                c.record_stream(self._xs)
            m += take
            if take == n:
                self._backlog.pop(0)
            else:
                self._backlog[0] = c[take:]
            self._backlog_plies -= take
        return m, 0

    _whole_games = staticmethod(RecordGatherer._whole_games)

    def _issue(self, j, plan, sz):
        """ONE collective of ``sz`` bytes per rank: ``[header 64 B | staged records]``, the smallest class that holds every rank's."""
        recv = self._recv2[j & 1][: self.world * sz]
        m, abort, games = self._staged
        with (torch.cuda.stream(self._xs) if self._xs is not None else _NullCtx()):
            if self._xs is not None and self._consumed_ev[j & 1] is not None:   # the consumer of exchange j - 2 read this receive slot
                self._xs.wait_event(self._consumed_ev[j & 1])
            hdr = torch.tensor([m, self._backlog_plies, 1 if self._closing else 0, games, abort, j, self.moves_posted, sz], dtype=torch.int64)
            self._send[:self.HEADER].view(torch.int64).copy_(hdr)       # (blocking on the side stream only)
            self._work = dist.all_gather_into_tensor(recv, self._send[:sz], group=self.group, async_op=True)
        self._inflight = (j, recv, sz, plan)
        self._staged = None
        self.plies_sent += m
        self.bytes_sent += sz
        self.last_slot_bytes = sz
        self.issued += 1
        self.collectives += 1

    def _payload_of(self, buf, k: int = 0, sz: int | None = None):
        sz = self.slot_bytes if sz is None else sz
        base = k * sz + self.HEADER
        n = (sz - self.HEADER) // REC_BYTES
        return buf[base: base + n * REC_BYTES].view(n, REC_BYTES)

    def _complete(self):
        """The exchange in flight has finished (or: wait for it): check the headers against the announcements, hand over the union."""
        j, recv, sz, plan = self._inflight
        work, self._work, self._inflight = self._work, None, None
        # the headers are read on the SIDE stream (which waits for the collective's stream only): the host does not wait for the
        # simulation steps queued on the main stream. gloo: a host wait.
        with (torch.cuda.stream(self._xs) if self._xs is not None else _NullCtx()):
            work.wait()
            heads = recv.view(self.world, sz)[:, :self.HEADER].contiguous().view(torch.int64).view(self.world, 8).cpu()
        if self._xs is not None:   # consumers of the union are enqueued on the current stream: behind the collective
            torch.cuda.current_stream(self.device).wait_stream(self._xs)
        self.completed += 1
        got = [[int(heads[k, 5]), int(heads[k, 0]), int(heads[k, 7])] for k in range(self.world)]
        want = [[j, plan[k][0], sz] for k in range(self.world)]
        if got != want:
            raise RuntimeError(f"exchange {j}: what arrived (index, plies, slot bytes per rank) {got} is not what was announced {want} "
                               f"(a collective was issued on this group outside the exchange?)")
        segs = []
        self.rows_per_rank = [plan[k][0] for k in range(self.world)]
        for k in range(self.world):
            if self.rows_per_rank[k]:
                segs.append(self._payload_of(recv, k, sz)[:self.rows_per_rank[k]])
        union = segs[0] if len(segs) == 1 else torch.cat(segs)
        self._all_closed = False      # somebody sent something: the closing exchange is the (virtual) one after it
        self.user_sum = sum(p[4] for p in plan)
        self._handed_over = j & 1
        return Exchanged(union, self.user_sum, list(self.rows_per_rank), j)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class TupleGatherer(_FusedGather):
    """All-gather of finished (state, pi, z) rows in their DENSE form (round 2's wire format; kept for consumers that
    hold dense rows only -- self-play exchanges compact records through :class:`RecordGatherer`).

    Send buffer of every rank (uint8, fixed size, allocated once):
        [ header 64 B | states cap x 21,420 B | pi cap x 8,344 B | z cap x 4 B ]
    One collective and one host sync per round of ``capacity_rows`` rows; more rows take further rounds of the same
    shape. Received rows are returned as one rank-major copy out of the receive buffer (no per-round clones).

    Bytes on the wire per round and rank: 64 + capacity_rows x 29,768 sent, (world - 1) x that received.
    """

    S_BYTES = STATE_ELEMS * 2
    P_BYTES = NMOVES * 4
    ROW_BYTES = STATE_ELEMS * 2 + NMOVES * 4 + 4

    def __init__(self, capacity_rows: int, device, group=None, always_collective: bool = False):
        self.cap = int(capacity_rows)
        if self.cap <= 0:
            raise ValueError("capacity_rows must be positive")
        c = self.cap
        self._o_s = self.HEADER
        self._o_p = self._o_s + c * self.S_BYTES
        self._o_z = self._o_p + c * self.P_BYTES
        super().__init__(self._o_z + c * 4 - self.HEADER, device, group, always_collective)

    def _sections(self, buf, k):
        base = k * self.slot_bytes
        c = self.cap
        s = buf[base + self._o_s: base + self._o_s + c * self.S_BYTES].view(c, self.S_BYTES)
        p = buf[base + self._o_p: base + self._o_p + c * self.P_BYTES].view(torch.float32).view(c, NMOVES)
        z = buf[base + self._o_z: base + self._o_z + c * 4].view(torch.float32)
        return s, p, z

    def gather(self, states: torch.Tensor, pi: torch.Tensor, z: torch.Tensor, more: bool = False, user: int = 0):
        """Every rank passes its new rows (possibly zero); every rank gets all rows in rank order.

        ``more``: this rank will call gather() again in the same exchange loop (it holds further harvest chunks);
        ``self.any_more`` is True afterwards if any rank said so. ``user`` is summed over ranks into
        ``self.user_sum`` (finished-game counts ride along instead of needing their own all-reduce)."""
        n = int(states.shape[0])
        self.rounds = self.collectives = 0
        if self._solo():
            self.any_more, self.user_sum, self.rows_per_rank = bool(more), int(user), [n]
            return states, pi, z
        sbits = states.contiguous().reshape(n, STATE_ELEMS).view(torch.uint8) if n else None
        outs = [[] for _ in range(self.world)]
        totals = [0] * self.world
        self.any_more, self.user_sum = False, 0
        ss, sp, sz = self._sections(self._send, 0)
        lo = 0
        while True:
            m = min(n - lo, self.cap)
            if m:
                ss[:m].copy_(sbits[lo:lo + m])
                sp[:m].copy_(pi[lo:lo + m])
                sz[:m].copy_(z[lo:lo + m])
            heads = self._round(m, n - lo - m, more, user if self.rounds == 0 else 0)
            again = bool((heads[:, 1] > 0).any())  # some rank still holds rows: another round of the same shape follows
            for k in range(self.world):
                mk = int(heads[k, 0])
                if mk:
                    s_k, p_k, z_k = self._sections(self._recv, k)
                    # the next round overwrites the receive buffer: only then are the rows copied out here
                    outs[k].append((s_k[:mk].clone(), p_k[:mk].clone(), z_k[:mk].clone()) if again else (s_k[:mk], p_k[:mk], z_k[:mk]))
                    totals[k] += mk
            lo += m
            if not again:
                break
        self.rows_per_rank = totals
        flat = [t for k in range(self.world) for t in outs[k]]
        if not flat:
            return states.reshape(0, 17, 7, 10, 9), pi[:0], z[:0]
        S = torch.cat([t[0] for t in flat]).view(torch.float16).reshape(-1, 17, 7, 10, 9)
        P = torch.cat([t[1] for t in flat])
        Z = torch.cat([t[2] for t in flat])
        return S, P, Z


class ReplayBuffer:
    """Fixed-capacity ring of training rows resident in HBM (sized for 288 GB: 1M rows = 30 GB)."""

    def __init__(self, capacity_rows: int, device):
        d = torch.device(device)
        self.cap = int(capacity_rows)
        self.states = torch.zeros((self.cap, 17, 7, 10, 9), dtype=torch.float16, device=d)
        self.pi = torch.zeros((self.cap, NMOVES), dtype=torch.float32, device=d)
        self.z = torch.zeros((self.cap,), dtype=torch.float32, device=d)
        self.size = 0
        self.head = 0
        self.total = 0

    def append(self, states, pi, z):
        n = int(states.shape[0])
        if n == 0:
            return
        if n > self.cap:
            states, pi, z = states[-self.cap:], pi[-self.cap:], z[-self.cap:]
            n = self.cap
        first = min(n, self.cap - self.head)
        self.states[self.head:self.head + first].copy_(states[:first])
        self.pi[self.head:self.head + first].copy_(pi[:first])
        self.z[self.head:self.head + first].copy_(z[:first])
        if n > first:
            r = n - first
            self.states[:r].copy_(states[first:])
            self.pi[:r].copy_(pi[first:])
            self.z[:r].copy_(z[first:])
        self.head = (self.head + n) % self.cap
        self.size = min(self.cap, self.size + n)
        self.total += n

    def append_records(self, records: torch.Tensor, flags: int = 0, plane_of_type=None, bad=None) -> int:
        """Expand compact ply records (uint8 [P, 880], whole games; :class:`RecordGatherer`'s output) straight INTO the ring:
        ``ccz_expand_records`` writes the dense rows at (head + i) % capacity, no intermediate copy. Returns the rows added."""
        from .engine import expand_records, game_aligned_chunks, rows_of_records
        mul = rows_of_records(1, flags)
        if rows_of_records(int(records.shape[0]), flags) > self.cap:  # more than the ring holds: game by game, the ring wraps
            if self.cap < mul:
                raise ValueError("replay ring smaller than one ply's rows")
            return sum(self.append_records(part, flags, plane_of_type, bad) for part in game_aligned_chunks(records, self.cap // mul))
        P = int(records.shape[0])
        n = P * mul
        if n == 0:
            return 0
        if n > self.cap:
            raise ValueError(f"one game of {n} rows exceeds the replay ring ({self.cap} rows)")
        rec = records if records.device == self.states.device else records.to(self.states.device, non_blocking=True)
        expand_records(rec.contiguous(), flags, plane_of_type, out=(self.states, self.pi, self.z), head_row=self.head, bad=bad)
        self.head = (self.head + n) % self.cap
        self.size = min(self.cap, self.size + n)
        self.total += n
        return n

    def sample(self, batch: int, generator=None):
        idx = torch.randint(0, self.size, (batch,), device=self.states.device, generator=generator)
        return self.states[idx], self.pi[idx], self.z[idx]


def _flat_bytes(tensors, device):
    """The tensors' memory as ONE uint8 buffer on ``device`` (any mix of dtypes), and the (offset, nbytes) of each."""
    spans, off = [], 0
    for t in tensors:
        nb = t.numel() * t.element_size()
        spans.append((off, nb))
        off += -(-nb // 16) * 16
    flat = torch.empty((off,), dtype=torch.uint8, device=device)
    return flat, spans


def broadcast_model(policy_value_net, src: int = 0, group=None, what: str = "state"):
    """Model hot-reload across ranks as ONE broadcast of one flat byte buffer (round 3 issued one collective per
    tensor, ~500 of them). The reference's collector loads its model once per process and never refreshes it (collect.py:49).

    ``what="state"`` (default): every parameter and buffer of the fp32 ``Net`` (204 MB at 40 x 256), bit-exact; each rank then
    rebuilds its fp16 inference copy -- afterwards any rank can ``save_model`` or train. ``what="inference"``: only the BN-folded
    fp16 inference copy (102 MB), written IN PLACE into the receivers' copy (same device addresses, so captured hipGraphs stay
    valid); for ranks that only ever evaluate. Either way ``weights_version`` moves, which empties evaluation caches keyed to it."""
    if what not in ("state", "inference"):
        raise ValueError("what must be 'state' or 'inference'")
    multi = dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if multi else 0
    if what == "state":
        net = policy_value_net.policy_value_net
        tensors = [t.data for t in list(net.parameters()) + list(net.buffers())]
    else:
        if getattr(policy_value_net, "_infer", None) is None:
            policy_value_net.refresh_inference_copy()
        inf = policy_value_net._infer
        derived = inf.derived_parameter_names()   # packed copies: re-derived locally (InferenceNet.repack_derived rebuilds this very set), not sent
        named = dict(inf.named_parameters())
        if not derived <= set(named):
            raise RuntimeError(f"InferenceNet.derived_parameter_names() lists unknown parameters: {sorted(derived - set(named))}")
        tensors = [p.data for n, p in named.items() if n not in derived]
    if multi and tensors:
        dev = tensors[0].device
        if dist.get_backend(group) == "gloo":
            dev = torch.device("cpu")
        with torch.no_grad():
            flat, spans = _flat_bytes(tensors, dev)
            if rank == src:
                for t, (o, nb) in zip(tensors, spans):
                    flat[o:o + nb].copy_(t.contiguous().reshape(-1).view(torch.uint8))
            dist.broadcast(flat, src=src, group=group)
            if rank != src:
                for t, (o, nb) in zip(tensors, spans):
                    t.copy_(flat[o:o + nb].view(t.dtype).view(t.shape))
    if what == "state":
        policy_value_net._fp32_stale = False
        if hasattr(policy_value_net, "refresh_inference_copy"):
            policy_value_net.refresh_inference_copy()
    else:
        inf = policy_value_net._infer
        if multi and rank != src:
            inf.repack_derived()
            # the receiver's fp32 net was NOT sent: refresh_inference_copy / save_model / training from it would revert the reload
            policy_value_net._fp32_stale = True
        policy_value_net._graph = None
        policy_value_net.weights_version += 1
    return policy_value_net
