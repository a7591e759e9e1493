"""Multi-GPU exchange: all-gather of finished (state, pi, z) rows.

Games shard embarrassingly over ranks (one process per GPU, own engine, RNG streams keyed by the
global board id); the ONLY collective is this all-gather, backend ``nccl`` (= RCCL over xGMI) on
GPUs and ``gloo`` in the CPU tests. Buffers are padded to a fixed capacity so that every rank issues
the same three collectives per exchange regardless of how many games finished where; volume is ~1
MB/s/GPU at target throughput, so latency (not xGMI bandwidth) is what the fixed shape buys back.
The reference has no counterpart: its "exchange" is N collector processes appending to one HDF5
file (collect.py:146-167); parity = the union of shards equals what N collectors would append.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

STATE_ELEMS = 17 * 7 * 10 * 9
NMOVES = 2086


class TupleGatherer:
    def __init__(self, capacity_rows: int, device, group=None):
        self.cap = int(capacity_rows)
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        d = self.device
        self._s = torch.zeros((self.cap, STATE_ELEMS * 2), dtype=torch.uint8, device=d)  # fp16 bytes (gloo-safe)
        self._p = torch.zeros((self.cap, NMOVES), dtype=torch.float32, device=d)
        self._z = torch.zeros((self.cap,), dtype=torch.float32, device=d)
        self._S = torch.zeros((self.world * self.cap, STATE_ELEMS * 2), dtype=torch.uint8, device=d)
        self._P = torch.zeros((self.world * self.cap, NMOVES), dtype=torch.float32, device=d)
        self._Z = torch.zeros((self.world * self.cap,), dtype=torch.float32, device=d)
        self._cnt = torch.zeros((1,), dtype=torch.int64, device=d)
        self._cnts = torch.zeros((self.world,), dtype=torch.int64, device=d)

    def gather(self, states: torch.Tensor, pi: torch.Tensor, z: torch.Tensor):
        """Every rank passes its new rows (possibly zero); every rank gets all rows in rank order."""
        n = int(states.shape[0])
        if self.world == 1:
            return states, pi, z
        self._cnt[0] = n
        dist.all_gather_into_tensor(self._cnts, self._cnt, group=self.group)
        counts = self._cnts.tolist()
        rounds = max(1, -(-max(counts) // self.cap))
        outs_s, outs_p, outs_z = [], [], []
        sbits = states.contiguous().reshape(n, STATE_ELEMS).view(torch.uint8)
        for r in range(rounds):
            lo = min(n, r * self.cap)
            hi = min(n, (r + 1) * self.cap)
            m = hi - lo
            if m:
                self._s[:m].copy_(sbits[lo:hi])
                self._p[:m].copy_(pi[lo:hi])
                self._z[:m].copy_(z[lo:hi])
            dist.all_gather_into_tensor(self._S, self._s, group=self.group)
            dist.all_gather_into_tensor(self._P, self._p, group=self.group)
            dist.all_gather_into_tensor(self._Z, self._z, group=self.group)
            for k, c in enumerate(counts):
                mk = min(c, (r + 1) * self.cap) - min(c, r * self.cap)
                if mk > 0:
                    outs_s.append(self._S[k * self.cap:k * self.cap + mk].clone())
                    outs_p.append(self._P[k * self.cap:k * self.cap + mk].clone())
                    outs_z.append(self._Z[k * self.cap:k * self.cap + mk].clone())
        if not outs_s:
            e = states.reshape(0, 17, 7, 10, 9)
            return e, pi[:0], z[:0]
        # rows arrive round-major; restore rank-major order
        order = []
        idx = 0
        per = {}
        for r in range(rounds):
            for k, c in enumerate(counts):
                mk = min(c, (r + 1) * self.cap) - min(c, r * self.cap)
                if mk > 0:
                    per.setdefault(k, []).append(idx)
                    idx += 1
        for k in sorted(per):
            order.extend(per[k])
        S = torch.cat([outs_s[i] for i in order]).view(torch.float16).reshape(-1, 17, 7, 10, 9)
        P = torch.cat([outs_p[i] for i in order])
        Z = torch.cat([outs_z[i] for i in order])
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
