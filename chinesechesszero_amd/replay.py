"""Multi-GPU exchange: all-gather of finished (state, pi, z) rows.

Games shard embarrassingly over ranks (one process per GPU, own engine, RNG streams keyed by the
global board id); the ONLY collective is this all-gather, backend ``nccl`` (= RCCL over xGMI) on
GPUs and ``gloo`` in the CPU tests. One fused, fixed-shape buffer per rank (header with the counts + the
three row sections), so that every rank issues the same ONE collective per exchange regardless of how
many games finished where; volume is ~1 MB/s/GPU at target throughput, so latency (not xGMI bandwidth)
is what the fixed shape and the single collective buy back.
The reference has no counterpart: its "exchange" is N collector processes appending to one HDF5
file (collect.py:146-167); parity = the union of shards equals what N collectors would append.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

STATE_ELEMS = 17 * 7 * 10 * 9
NMOVES = 2086


class TupleGatherer:
    """All-gather of finished (state, pi, z) rows: ONE collective per exchange.

    Send buffer of every rank (uint8, fixed size, allocated once):
        [ header 64 B | states cap x 21,420 B | pi cap x 8,344 B | z cap x 4 B ]
    header = int64 x 8: rows in this round, rows this rank still holds after it, ``more`` flag of the caller,
    ``user`` counter (finished games), 4 spare. Counts therefore travel inside the same
    ``all_gather_into_tensor`` as the rows; after it ONE small device-to-host copy (world x 64 B) tells every
    rank how many rows each section holds. When every rank has at most ``capacity_rows`` rows (the norm: ~8 k
    rows per move and rank at 4096 boards) an exchange is exactly one collective and one host sync; more rows
    take further rounds of the same shape. Received rows are returned as one rank-major copy out of the
    receive buffer (no per-round clones).

    Bytes on the wire per exchange and rank: 64 + capacity_rows x 29,768 sent, (world - 1) x that received
    (padded: the shape is fixed so that every rank issues the same collective whatever finished where).
    """

    HEADER = 64
    S_BYTES = STATE_ELEMS * 2
    P_BYTES = NMOVES * 4
    ROW_BYTES = STATE_ELEMS * 2 + NMOVES * 4 + 4

    def __init__(self, capacity_rows: int, device, group=None, always_collective: bool = False):
        """``always_collective``: issue the collective even in a group of one (exercises the RCCL path on a single GPU)."""
        self.always_collective = bool(always_collective)
        self.cap = int(capacity_rows)
        if self.cap <= 0:
            raise ValueError("capacity_rows must be positive")
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        c = self.cap
        self._o_s = self.HEADER
        self._o_p = self._o_s + c * self.S_BYTES
        self._o_z = self._o_p + c * self.P_BYTES
        self.slot_bytes = -(-(self._o_z + c * 4) // 64) * 64
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
        self.rows_per_rank: list[int] = []

    def bytes_per_exchange(self) -> int:
        """Bytes one rank sends in one round (it receives (world - 1) x this)."""
        return self.slot_bytes

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
        if self.world == 1 and not (self.always_collective and dist.is_initialized()):
            self.any_more, self.user_sum, self.rows_per_rank = bool(more), int(user), [n]
            return states, pi, z
        sbits = states.contiguous().reshape(n, STATE_ELEMS).view(torch.uint8) if n else None
        outs = [[] for _ in range(self.world)]
        totals = [0] * self.world
        self.any_more, self.user_sum = False, 0
        ss, sp, sz = self._sections(self._send, 0)
        hdr = self._send[:self.HEADER].view(torch.int64)
        lo = 0
        while True:
            m = min(n - lo, self.cap)
            self._hdr_host[0] = m
            self._hdr_host[1] = n - lo - m
            self._hdr_host[2] = 1 if more else 0
            self._hdr_host[3] = int(user) if self.rounds == 0 else 0
            hdr.copy_(self._hdr_host, non_blocking=True)
            if m:
                ss[:m].copy_(sbits[lo:lo + m])
                sp[:m].copy_(pi[lo:lo + m])
                sz[:m].copy_(z[lo:lo + m])
            dist.all_gather_into_tensor(self._recv, self._send, group=self.group)
            self.collectives += 1
            self.rounds += 1
            heads = self._recv.view(self.world, self.slot_bytes)[:, :self.HEADER].contiguous().view(torch.int64).view(self.world, 8).cpu()  # the one host sync
            again = bool((heads[:, 1] > 0).any())  # some rank still holds rows: another round of the same shape follows
            for k in range(self.world):
                mk, _, mo, us = (int(v) for v in heads[k, :4])
                self.any_more |= bool(mo)
                self.user_sum += us
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

    def sample(self, batch: int, generator=None):
        idx = torch.randint(0, self.size, (batch,), device=self.states.device, generator=generator)
        return self.states[idx], self.pi[idx], self.z[idx]


def broadcast_model(policy_value_net, src: int = 0, group=None):
    """Model hot-reload across ranks: broadcast every parameter and buffer of the
    ``PolicyValueNet`` from ``src`` (the trainer's rank) and rebuild the inference copy. The reference's
    collector loads its model once per process and never refreshes it (collect.py:49)."""
    net = policy_value_net.policy_value_net
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        with torch.no_grad():
            for t in list(net.parameters()) + list(net.buffers()):
                dist.broadcast(t.data, src=src, group=group)
    if hasattr(policy_value_net, "refresh_inference_copy"):
        policy_value_net.refresh_inference_copy()
    return policy_value_net
