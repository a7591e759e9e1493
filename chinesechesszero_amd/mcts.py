"""Host mirror of reference mcts.py: ``MCTS`` and ``MCTS_AI`` with the reference's call surface,
backed by the HIP lockstep engine (one board per player object).

Same constructor arguments, methods and return conventions as reference mcts.py:81-233, so
``Game.start_self_play(player)`` and the reference's own call sites read unchanged. The tree lives
in device memory; ``Node`` objects do not exist on the host. pi and the move choice follow the
reference's NumPy code literally (softmax(1/temp*log(N+1e-10)), global ``np.random`` Dirichlet +
choice), so under ``np.random.seed`` a single game reproduces the reference's moves.

Evaluator forms accepted as ``policy_value_fn``:
  * a callable with attribute ``batched = True`` taking the fp16 leaf tensor [B,17,7,10,9] on the
    device and returning ``(prob float32 [B,2086], value float32 [B])`` (the fast path; e.g.
    ``PolicyValueNet.evaluate_leaves``),
  * ``PolicyValueNet.policy_value_fn`` (bound method): routed to ``evaluate_leaves`` of the same object,
  * any reference-style ``f(board) -> (iterable[(move_id, prob)], value)`` (compatibility path: the
    leaf is handed to ``f`` as a light ``LeafView``; one device round trip per playout).
"""
from __future__ import annotations

import numpy as np
import torch

from .engine import SelfPlayEngine
from .parameters import ALPHA, C_PUCT, EPS
from .tools import softmax


class LeafView:
    """What a reference-style evaluator may ask of the leaf board (net.py:151-177)."""

    def __init__(self, squares: np.ndarray, turn: bool, ids: list[int]):
        self._sq = squares
        self.turn = turn
        self._ids = ids

    def squares(self):
        return self._sq.copy()

    def legal_ids(self):
        return list(self._ids)

    @property
    def legal_moves(self):
        from .game import Move
        return [Move.from_id(i) for i in self._ids]

    def piece_at(self, square):
        from .game import Piece
        pc = int(self._sq[square])
        return Piece(pc & 7, not bool(pc & 8)) if pc else None


def _planes_to_squares(planes: np.ndarray):
    from . import tools
    p = planes.reshape(17, 7, 90)
    types = np.zeros(7, np.int64)
    for t in range(1, 8):
        types[tools.PLANE_OF_TYPE[t]] = t   # channel -> piece type
    types = types.reshape(7, 1)
    return ((p[7] * types).sum(0) + (p[15] * (types + 8)).sum(0)).astype(np.uint8), bool(p[16, 0, 0] > 0)


SCOUTS = 10  # scout slots of a one-game search (ScoutedSearch): 1 + 10 = 11 rows = 990 pixels = 16 x 16 blocks of k_conv3x3_small = ONE round of the 256
             # CUs, at the price of one row (7.0 against 6.5 us per tower layer), and an evaluator call answers 4.2 simulations on average: 6,000
             # sims/s against 5,626 with 7 and 3,494 with 15 (a second round of blocks); profiles/r06_single_board.json. env CCZ_SCOUTS, 0 = off


class MCTS:
    def __init__(self, policy_value_fn, c_puct=5, n_playout=10000, device: int = 0, seed: int = 0, scouts: int | None = None):
        """``scouts``: scout slots of the one-game search (``selfplay.ScoutedSearch``; same visit counts, fewer evaluator calls);
        None = ``SCOUTS`` (env ``CCZ_SCOUTS``) when the evaluator is a batched one that returns logits, else 0."""
        self.policy = policy_value_fn
        self.c_puct = c_puct
        self.n_playout = n_playout
        self.red_history = None
        self.black_history = None
        self._device = device
        self._seed = seed
        self._engine: SelfPlayEngine | None = None
        self._synced: list[int] | None = None   # move ids pushed on the engine since its start position
        self._start = None
        self._discard = False
        self._graph = None
        self.use_graph = True   # single-board play is launch-bound: replay (evaluator, k_step) as one hipGraph
        owner = getattr(policy_value_fn, "__self__", None)
        if getattr(policy_value_fn, "batched", False):
            self._batched = policy_value_fn
        elif owner is not None and hasattr(owner, "evaluate_leaves") and getattr(policy_value_fn, "__name__", "") == "policy_value_fn":
            self._batched = getattr(owner, "evaluate_leaves_logits", owner.evaluate_leaves)
        else:
            self._batched = None
        import os
        want = int(os.environ.get("CCZ_SCOUTS", SCOUTS)) if scouts is None else int(scouts)
        ok = self._batched is not None and getattr(self._batched, "returns_logits", False)
        self.scouts = max(0, min(want, 63)) if ok else 0
        self._scouted = None

    # ---- engine management -----------------------------------------------------------------------
    def _ensure_engine(self):
        if self._engine is None:
            # board 0 is the game; the scout slots behind it (no tree, no game of their own) carry the leaves board 0 will ask for next
            # through a small evaluation cache (2^16 positions, 34 MB)
            self._engine = SelfPlayEngine(1 + self.scouts, n_playout=max(1, self.n_playout), c_puct=self.c_puct, eps=EPS, alpha=ALPHA,
                                          device=self._device, seed=self._seed, mirror=True, eval_cache_log2=16 if self.scouts else 0,
                                          strict=True)   # the reference's tree and game are unbounded: a prune or an adjudication raises here
            if self.scouts:
                self._engine.set_scouts(self.scouts)
        return self._engine

    def _forced(self, mid: int):
        f = np.full(self._engine.B, -1, np.int32)   # (the scout slots are not played: the simulator kernels run on board 0 only)
        f[0] = int(mid)
        return f

    def _sync_root(self, board):
        """Make the engine's root the position of ``board`` (same move history => same repetition state)."""
        e = self._ensure_engine()
        ids = [m.id for m in board.move_stack]
        start = board._start
        same_start = self._start is not None and np.array_equal(self._start[0], start[0]) and self._start[1:] == start[1:]
        if not same_start or self._synced is None or ids[:len(self._synced)] != self._synced:
            e.set_position(0, start[0], 1 if start[1] else 0, start[2])
            self._start = (start[0].copy(), start[1], start[2])
            self._synced = []
            self._discard = False
        new = ids[len(self._synced):]
        for mid in new:
            e.finish_move(forced_moves=self._forced(mid), keep_tree=not self._discard)
            self._synced.append(mid)
        if self._discard and not new:
            self._reset_tree_keep_position(board)
        self._discard = False
        assert np.array_equal(e.root_positions()[0], board.squares()), "engine root out of sync with the board"

    def _reset_tree_keep_position(self, board):
        # Node(None, 1.0) (mcts.py:176-178): one launch; position, history chain and clocks stay on the engine
        self._engine.reset_tree()
        self._discard = False

    # ---- reference surface ------------------------------------------------------------------------
    def playout(self, board=None, red_states=None, black_states=None):
        """One simulation (mcts.py:101-129) of the engine's root."""
        e = self._ensure_engine()
        leaf = e.select_leaves()
        if self._batched is not None:
            prob, value = self._batched(leaf)
            if getattr(self._batched, "returns_logits", False):
                e.expand_backup_logits(prob, value)
                return
        else:
            info = e.leaf_info()
            sq, turn = _planes_to_squares(leaf[0].float().cpu().numpy())
            k = int(info["k"][0])
            view = LeafView(sq, turn, info["ids"][0][:k].astype(np.int64).tolist())
            act_probs, leaf_value = self.policy(view, red_states, black_states)
            p = np.zeros((1, 2086), np.float32)
            for a, pr in act_probs:
                p[0, int(a)] = np.float32(pr)
            prob = torch.from_numpy(p).to(e.device)
            value = torch.from_numpy(np.asarray(leaf_value, np.float32).reshape(1)).to(e.device)
        e.expand_backup(prob, value)

    def get_move_probs(self, board, temp=1e-3, red_states=None, black_states=None, on_playout=None):
        """mcts.py:131-166: run n_playout simulations, return (acts, probs) over the root's children."""
        self.red_history = red_states
        self.black_history = black_states
        self._sync_root(board)
        interval = max(1, self.n_playout // 100)
        acc = 0
        e = self._engine
        fused = self._batched is not None
        logits = fused and getattr(self._batched, "returns_logits", False)
        if fused and self.scouts:
            # one game at a time with scout slots: the evaluator runs only when board 0's leaf is not in the table yet, on 1 + scouts rows
            if self._scouted is None:
                from .selfplay import ScoutedSearch
                self._scouted = ScoutedSearch(e, self._batched, use_graph=self.use_graph)
            self._scouted.begin_move()
        elif fused:
            # launch sequence of a move: select, (evaluator, fused step) x (n-1), evaluator, expand_backup
            leaf = e.select_leaves()
            if self.use_graph and self._graph is None and getattr(self._batched, "graph_safe", False) and e.device.type == "cuda":
                from .selfplay import GraphedStep
                self._graph = GraphedStep(e, self._batched)

        def report(last):
            nonlocal acc
            if on_playout is not None and (acc >= interval or last):   # mcts.py:154-160
                try:
                    on_playout(acc)
                except Exception:
                    pass
                acc = 0

        if fused and self.scouts and self._scouted.device_loop:
            # simulations that find their leaf in the table repeat on the device (ccz_scouted_run): the host sees the search when the
            # evaluator has to run -- and at the playouts on_playout is due at, the same ones as in the loop below
            left = self.n_playout
            while left > 0:
                done = self._scouted.run(left, left if on_playout is None else min(left, interval - acc))
                left -= done
                acc += done
                report(left == 0)
            n_host = 0
        else:
            n_host = self.n_playout
        for i in range(n_host):
            if not fused:
                self.playout(board, red_states, black_states)
            elif self.scouts:
                self._scouted.simulate(last=i + 1 == self.n_playout)
            elif i + 1 < self.n_playout:
                if self._graph is not None:
                    self._graph.replay()
                else:
                    leaf = (e.step_logits if logits else e.step)(*self._batched(leaf))
            else:
                (e.expand_backup_logits if logits else e.expand_backup)(*self._batched(leaf))
            acc += 1
            report(i == self.n_playout - 1)
        rc = self._engine.root_children()
        self._engine.check_healthy()
        k = int(rc["k"][0])
        acts = tuple(int(a) for a in rc["acts"][0][:k])
        visits = rc["visits"][0][:k].astype(np.int64)
        act_probs = softmax(1.0 / temp * np.log(np.array(visits) + 1e-10))
        return acts, act_probs

    def update_with_move(self, last_move):
        """mcts.py:168-178: keep the subtree of ``last_move``; -1 (or an unknown move) starts a fresh root."""
        if last_move == -1 or self._engine is None or self._synced is None:
            self._discard = True
            return
        self._engine.finish_move(forced_moves=self._forced(last_move), keep_tree=True)
        self._synced.append(int(last_move))

    def root_children(self):
        rc = self._ensure_engine().root_children()
        k = int(rc["k"][0])
        return {key: (val[0][:k] if val.ndim == 2 else val[0]) for key, val in rc.items()}


class MCTS_AI:
    """reference mcts.py:181-233"""

    def __init__(self, policy_value_fn, c_puct=5, n_playout=2000, is_selfplay=False, device: int = 0, seed: int = 0):
        self.mcts = MCTS(policy_value_fn, c_puct, n_playout, device=device, seed=seed)
        self.is_selfplay = is_selfplay
        self.agent = "AI"

    def set_player_idx(self, p):
        self.player = p

    def reset_player(self):
        self.mcts.update_with_move(-1)

    def get_action(self, board, temp=1e-3, return_prob=False, on_playout=None):
        move_probs = np.zeros(2086)
        acts, probs = self.mcts.get_move_probs(board, temp, on_playout=on_playout)
        move_probs[list(acts)] = probs
        if self.is_selfplay:
            # Dirichlet noise on the sampling distribution only (mcts.py:216-221)
            move = np.random.choice(acts, p=(1 - EPS) * probs + EPS * np.random.dirichlet(ALPHA * np.ones(len(probs))))
            self.mcts.update_with_move(move)
        else:
            move = np.random.choice(acts, p=probs)
            self.mcts.update_with_move(-1)
        if return_prob:
            return move, move_probs
        return move
