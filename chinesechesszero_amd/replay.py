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
        derived = {n for n, _ in inf.named_parameters() if "g16" in n}   # packed copies: re-derived locally, not sent
        tensors = [p.data for n, p in inf.named_parameters() if n not in derived]
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
        if hasattr(policy_value_net, "refresh_inference_copy"):
            policy_value_net.refresh_inference_copy()
    else:
        inf = policy_value_net._infer
        if multi and rank != src:
            inf.repack_derived()
        policy_value_net._graph = None
        policy_value_net.weights_version += 1
    return policy_value_net
