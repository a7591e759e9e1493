"""Deterministic stand-in evaluators for parity tests (test infrastructure, NOT the product).

The reference's evaluator boundary is ``policy_value_fn(board) -> (zip(ids, P[ids]), value)``
(reference net.py:151-205): P is a float32 vector over the whole 2086-move action space and the
value an ndarray(1,1) float32. The batched fp16 net is not bit-reproducible across batch sizes, so
parity tests inject (P, v) instead (SURVEY hard part 2). These evaluators are pure integer hashing
followed by IEEE-exact conversions -- no exp/log -- so every host computes the same bits.
"""
from __future__ import annotations

import numpy as np

NMOVES = 2086
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def position_hash(sq: np.ndarray, turn: np.ndarray, salt: int = 0) -> np.ndarray:
    """FNV-1a style hash of ``sq`` uint8[n,90] and ``turn`` [n] -> uint64[n]."""
    sq = np.atleast_2d(np.asarray(sq, dtype=np.uint8))
    turn = np.atleast_1d(np.asarray(turn)).astype(np.uint64)
    with np.errstate(over="ignore"):
        h = np.full(sq.shape[0], np.uint64(0xCBF29CE484222325) ^ np.uint64(salt), dtype=np.uint64)
        for s in range(90):
            h = (h ^ sq[:, s].astype(np.uint64)) * np.uint64(0x100000001B3)
        h = (h ^ turn) * np.uint64(0x100000001B3)
    return h


def hash_eval(sq, turn, salt: int = 0, scale: float = 1.0):
    """(P float32[n,2086], v float32[n]) from positions; bit-identical on every IEEE host.

    ``scale`` > 1 concentrates prior mass on fewer moves (capped at 1.0) to mimic a trained net;
    ``scale`` == 1 spreads mass over all 2086 ids like a random-init net.
    """
    h = position_hash(sq, turn, salt)
    ids = np.arange(NMOVES, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = _splitmix64(h[:, None] + ids[None, :] * np.uint64(0x9E3779B97F4A7C15))
        r = ((x >> np.uint64(52)) + np.uint64(1)).astype(np.int64)  # 1 .. 4096
        w = r * r * r * r                                             # <= 2^48, exact
        s = w.sum(axis=1, dtype=np.int64)                            # exact in int64
        p = np.minimum(1.0, float(scale) * (w.astype(np.float64) / s[:, None].astype(np.float64)))
        xv = _splitmix64(h ^ np.uint64(0xA5A5A5A5A5A5A5A5))
        v = ((xv >> np.uint64(40)).astype(np.int64) - (1 << 23)).astype(np.float64) / float(1 << 23)
    return p.astype(np.float32), v.astype(np.float32)


def uniform_eval(sq, turn):
    """Constant priors, zero value: every PUCT comparison is an exact tie (first-max order test)."""
    n = np.atleast_2d(sq).shape[0]
    return np.full((n, NMOVES), np.float32(1.0 / NMOVES), dtype=np.float32), np.zeros(n, dtype=np.float32)


EVALUATORS = {
    "hash": lambda sq, turn: hash_eval(sq, turn, salt=0, scale=1.0),
    "hash_sharp": lambda sq, turn: hash_eval(sq, turn, salt=7, scale=40.0),
    "uniform": uniform_eval,
}
