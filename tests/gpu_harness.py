"""Shared helpers for the -m gpu parity tests: drive the HIP engine and the CPU oracle in lockstep."""
from __future__ import annotations

import numpy as np
import torch

from oracle import OracleBoard, OracleMCTS
from oracle.evaluators import hash_eval, uniform_eval


def planes_to_squares(planes: np.ndarray, plane_of_type=None):
    """leaf input [B,17,7,10,9] (0/1) -> (squares uint8 [B,90], turn [B]); also checks the static zeros.
    ``plane_of_type``: the engine's piece-type -> channel table (default type-1)."""
    p = planes.reshape(planes.shape[0], 17, 7, 90)
    assert p[:, :7].sum() == 0 and p[:, 8:15].sum() == 0, "history groups must stay zero on the search path"
    types = np.arange(1, 8, dtype=np.int64)
    if plane_of_type is not None:
        for t in range(1, 8):
            types[int(plane_of_type[t])] = t
    types = types[None, :, None]
    red = (p[:, 7] * types).sum(axis=1)
    black = (p[:, 15] * (types + 8)).sum(axis=1)
    assert np.all((p[:, 7].sum(axis=1) + p[:, 15].sum(axis=1)) <= 1)
    turn_plane = p[:, 16].reshape(p.shape[0], -1)
    assert np.all((turn_plane == turn_plane[:, :1]).all(axis=1))
    return (red + black).astype(np.uint8), turn_plane[:, 0].astype(np.uint8)


def make_evaluator(kind: str, salts):
    """Batched evaluator on (squares, turn) with a per-board salt; returns (P [B,2086] f32, v [B] f32)."""
    salts = list(salts)

    def ev(sq, turn, rows=None):
        rows = range(len(sq)) if rows is None else rows
        P = np.zeros((len(sq), 2086), np.float32)
        V = np.zeros(len(sq), np.float32)
        for j, b in enumerate(rows):
            if kind == "uniform":
                p, v = uniform_eval(sq[j:j + 1], turn[j:j + 1])
            elif kind == "hash":
                p, v = hash_eval(sq[j:j + 1], turn[j:j + 1], salt=salts[b], scale=1.0)
            else:
                p, v = hash_eval(sq[j:j + 1], turn[j:j + 1], salt=salts[b], scale=40.0)
            P[j], V[j] = p[0], v[0]
        return P, V

    return ev


class Lockstep:
    """B engine boards next to B independent sequential oracles (reference mcts.py restated in C)."""

    def __init__(self, engine, boards, kind="hash", salts=None, c_puct=5, value_f16=False):
        self.e = engine
        self.B = engine.B
        self.boards = boards  # list[OracleBoard], root positions (also set on the engine by the caller)
        self.salts = list(range(self.B)) if salts is None else salts
        self.ev = make_evaluator(kind, self.salts)
        self.mcts = [OracleMCTS(None, c_puct=c_puct, n_playout=0, value_f16=value_f16) for _ in range(self.B)]

    def _after_select(self, check_leaf):
        """Leaf of every board is selected on the engine: select on the oracles too, compare, evaluate."""
        e = self.e
        planes = e.leaf_input.float().cpu().numpy()
        info = e.leaf_info()
        sq, turn = planes_to_squares(planes, getattr(e, "plane_of_type", None))
        P, V = self.ev(sq, turn)
        pending = []
        for b in range(self.B):
            if info["status"][b] == 3:
                pending.append(None)
                continue
            leaf, depth = self.mcts[b].select(self.boards[b])
            ids = leaf.legal_ids()
            if check_leaf:
                assert depth == info["depth"][b], (b, depth, info["depth"][b])
                assert np.array_equal(leaf.squares(), sq[b]), b
                assert int(leaf.turn) == int(turn[b])
                assert info["k"][b] == len(ids)
                assert info["ids"][b][: len(ids)].tolist() == ids
                end, tie = leaf.is_game_over(), leaf.is_tie()
                want = 0 if (not end and not tie) else (1 if (end and tie) else 2)
                assert info["status"][b] == want, (b, info["status"][b], want)
                assert np.array_equal(planes[b], leaf.leaf_planes())
            pending.append((leaf, ids))
        return P, V, pending

    def _oracle_backup(self, P, V, pending):
        for b, item in enumerate(pending):
            if item is not None:
                leaf, ids = item
                self.mcts[b].expand_backup(leaf, ids, P[b][ids], V[b])

    def step(self, check_leaf=True):
        e = self.e
        e.select_leaves()
        P, V, pending = self._after_select(check_leaf)
        self._oracle_backup(P, V, pending)
        e.expand_backup(torch.from_numpy(P).to(e.device), torch.from_numpy(V).to(e.device))

    def run_fused(self, n, check_leaf=True):
        """n simulations through the fused launch sequence: select, (evaluate, step) x (n-1), evaluate, expand_backup."""
        e = self.e
        e.select_leaves()
        for i in range(n):
            P, V, pending = self._after_select(check_leaf)
            self._oracle_backup(P, V, pending)
            tp, tv = torch.from_numpy(P).to(e.device), torch.from_numpy(V).to(e.device)
            if i + 1 < n:
                e.step(tp, tv)
            else:
                e.expand_backup(tp, tv)

    def compare_roots(self):
        rc = self.e.root_children()
        over = self.e.game_status()["over"]
        for b in range(self.B):
            if over[b]:
                continue  # a finished board has no live tree (its pool half is recycled by the global flip)
            acts, visits, q, prior = self.mcts[b].root_children()
            k = len(acts)
            assert rc["k"][b] == k, (b, rc["k"][b], k)
            assert np.array_equal(rc["acts"][b][:k], acts.astype(np.uint16)), b
            assert np.array_equal(rc["visits"][b][:k], visits), (b, rc["visits"][b][:k], visits)
            assert np.array_equal(rc["q"][b][:k].view(np.uint32), q.view(np.uint32)), b
            assert np.array_equal(rc["prior"][b][:k].view(np.uint32), prior.view(np.uint32)), b
            assert rc["root_visits"][b] == self.mcts[b].root_visits()
        return rc

    def play(self, moves):
        """Force `moves` (list of ids, -1 = skip board) on both sides with tree reuse."""
        self.e.finish_move(forced_moves=np.asarray(moves, np.int32))
        for b, m in enumerate(moves):
            if m >= 0:
                self.mcts[b].update_with_move(int(m))
                self.boards[b].push_id(int(m))


class SampleMirror:
    """A big lockstep batch with a SAMPLE of its boards mirrored on sequential oracles.

    The device evaluator runs on all B leaves; the rows of the sampled boards are copied to the host and the SAME
    float32 numbers are fed to the oracle's expansion/backup, so N / Q / P of the sampled trees must agree bit for bit
    with the engine's at full batch size (the batched evaluator itself is not what is being compared)."""

    def __init__(self, engine, sample, boards=None, c_puct=5, check_every=16):
        self.e = engine
        self.sample = [int(b) for b in sample]
        self.boards = boards if boards is not None else [OracleBoard() for _ in self.sample]
        self.mcts = [OracleMCTS(None, c_puct=c_puct, n_playout=0) for _ in self.sample]
        self.idx = torch.as_tensor(self.sample, device=engine.device, dtype=torch.long)
        self.check_every = check_every
        self.steps = 0

    def backup_on_oracles(self, prob, value):
        """Call after the evaluator, BEFORE the engine consumes (prob, value): the oracles select the same leaf and back the
        same numbers up."""
        e = self.e
        P = prob.index_select(0, self.idx).cpu().numpy()
        V = value.index_select(0, self.idx).cpu().numpy()
        info = e.leaf_info() if self.steps % self.check_every == 0 else None
        for j, b in enumerate(self.sample):
            leaf, depth = self.mcts[j].select(self.boards[j])
            ids = leaf.legal_ids()
            if info is not None and info["status"][b] != 3:
                assert depth == info["depth"][b] and info["k"][b] == len(ids), (b, depth, info["depth"][b])
                assert info["ids"][b][:len(ids)].tolist() == ids, b
                end, tie = leaf.is_game_over(), leaf.is_tie()
                assert info["status"][b] == (0 if (not end and not tie) else (1 if (end and tie) else 2)), b
            self.mcts[j].expand_backup(leaf, ids, P[j][ids], V[j])
        self.steps += 1

    def backup_on_oracles_compact(self, prior128, value, check=True):
        """The compact / planned evaluator boundary: ``prior128`` float32 [B,128] and ``value`` float32 [B] are what
        ``engine.leaf_priors()`` returned AFTER ``gather_priors[_planned]`` -- the numbers the tree is about to consume, whether they
        came from the evaluator's row of this board, from another board's row (a duplicate leaf of the same step) or from the
        evaluation cache. Entry i of a row belongs to the i-th legal id. Call before ``step_compact`` / ``expand_backup_compact``."""
        e = self.e
        info = e.leaf_info() if (check and self.steps % self.check_every == 0) else None
        status = info["status"] if info is not None else e.leaf_info()["status"]
        for j, b in enumerate(self.sample):
            if status[b] == 3:      # finished board: no pending leaf on the engine, nothing to follow
                continue
            leaf, depth = self.mcts[j].select(self.boards[j])
            ids = leaf.legal_ids()
            if info is not None:
                assert depth == info["depth"][b] and info["k"][b] == len(ids), (b, depth, info["depth"][b])
                assert info["ids"][b][:len(ids)].tolist() == ids, b
                end, tie = leaf.is_game_over(), leaf.is_tie()
                assert info["status"][b] == (0 if (not end and not tie) else (1 if (end and tie) else 2)), b
            self.mcts[j].expand_backup(leaf, ids, prior128[b][:len(ids)], value[b])
        self.steps += 1

    def compare_roots(self, rc=None):
        rc = self.e.root_children() if rc is None else rc
        for j, b in enumerate(self.sample):
            acts, visits, q, prior = self.mcts[j].root_children()
            k = len(acts)
            assert rc["k"][b] == k, (b, rc["k"][b], k)
            assert np.array_equal(rc["acts"][b][:k], acts.astype(np.uint16)), b
            assert np.array_equal(rc["visits"][b][:k], visits), (b, rc["visits"][b][:k], visits)
            assert np.array_equal(rc["q"][b][:k].view(np.uint32), q.view(np.uint32)), b
            assert np.array_equal(rc["prior"][b][:k].view(np.uint32), prior.view(np.uint32)), b
            assert rc["root_visits"][b] == self.mcts[j].root_visits(), b
        return rc

    def restarted(self, mask):
        """The engine restarted the boards whose mask byte is set (``engine.reset(mask)``): fresh game, fresh tree."""
        for j, b in enumerate(self.sample):
            if mask[b]:
                self.boards[j] = OracleBoard()
                self.mcts[j].update_with_move(-1)

    def played(self, moves, keep_tree=True):
        """The engine played ``moves`` (int array [B], -1 = none): follow on the oracles."""
        for j, b in enumerate(self.sample):
            m = int(moves[b])
            if m >= 0:
                self.mcts[j].update_with_move(m if keep_tree else -1)
                self.boards[j].push_id(m)
