"""Lockstep batched self-play: the GPU-native form of reference collect.py:133-143 + game.py:133-237.

The reference plays ONE game at a time, one batch-1 net call per playout. Here B games advance in
lockstep: per simulation ``select_leaves -> evaluator -> expand_backup`` for all boards (the
evaluator is PyTorch-ROCm on the same stream), per move ``finish_move``; finished games are
harvested into (state, pi, z) rows and restart immediately.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .engine import SelfPlayEngine
from .parameters import ALPHA, C_PUCT, EPS


class GraphedStep:
    """hipGraph of one inner iteration ``evaluator(leaf) -> ccz_step`` (capture launch-bound loops in graphs).

    At B = 4096 a step is 47 ms of GPU work and launches do not matter; at small B (single-board play, UCI)
    the ~130 launches of a batch-1 forward are host-bound, and replaying one captured graph per simulation
    removes that. The evaluator must be capture-safe (no host sync, static shapes): the MIOpen find step and
    allocator warm-up therefore run eagerly on a side stream first.
    """

    def __init__(self, engine: SelfPlayEngine, evaluator, warmup: int = 3, version_fn=None):
        """``version_fn() -> hashable``: what tells this object that the evaluator's weights changed (a re-capture follows).
        Default: the ``weights_version`` of the object the evaluator is a bound method of (``PolicyValueNet``). An evaluator
        wrapped in a lambda / partial / closure has no such owner: pass ``version_fn`` (e.g. ``lambda: pvn.weights_version``),
        otherwise the captured graph can never notice a weight change -- that case is logged once."""
        self.engine = engine
        self.evaluator = evaluator
        self.warmup = warmup
        self.captures = 0
        self.version_fn = version_fn
        if version_fn is None and not hasattr(getattr(evaluator, "__self__", None), "weights_version") \
                and not getattr(evaluator, "stateless", False) and getattr(evaluator, "__name__", "") != "uniform_evaluator":
            from .tools import log
            log("GraphedStep: the evaluator exposes no weights_version and no version_fn was given: the captured hipGraph will "
                "keep replaying the weights it was captured with (pass version_fn=... if they can change)", "WARNING")
        self._capture()

    def _weights_version(self):
        """The captured graph holds the device addresses of the evaluator's inference weights. ``PolicyValueNet`` bumps
        ``weights_version`` whenever that copy is rebuilt or invalidated (training step, hot reload): a replay against
        the old addresses would silently search with stale or freed weights."""
        if self.version_fn is not None:
            return self.version_fn()
        owner = getattr(self.evaluator, "__self__", None)
        return getattr(owner, "weights_version", None)

    def _capture(self):
        engine, evaluator = self.engine, self.evaluator
        dev = engine.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(self.warmup):  # results are discarded: the pending leaf is only evaluated, never stepped
                evaluator(engine.leaf_input)  # (also rebuilds a stale inference copy OUTSIDE the capture)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.version = self._weights_version()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            prob, value = evaluator(engine.leaf_input)
            if getattr(evaluator, "returns_logits", False):
                engine.step_logits(prob, value)
            else:
                engine.step(prob, value)
        # capture does not execute: nothing has been applied to the trees yet
        self.captures += 1

    def replay(self):
        if self._weights_version() != self.version:
            self._capture()  # weights changed since the capture (train_step / refresh_inference_copy / broadcast_model)
        self.graph.replay()


class ScoutedSearch:
    """One game at a time with SCOUT SLOTS (``include/cczero.h`` ccz_scout; round 6). The reference's first-maximum rule makes the
    children of a node first-visited in ``legal_moves`` order (mcts.py:47-48,59-61), so the leaves the next simulations will ask
    for are known: the pending leaf's next siblings. The engine has ``1 + scouts`` boards; board 0 is searched, the scout slots
    carry those siblings through the evaluation cache. Per simulation: ``ccz_step_compact`` (expand + backup + select) ->
    ``ccz_scout`` -> probe + plan -> the HOST reads whether board 0's leaf is already in the table; only if it is not, the
    evaluator runs -- once, on all ``1 + scouts`` rows (latency-bound: 11 rows -- one round of blocks -- cost what 1 costs) -- and the scouts' results are
    stored for the simulations to come. Two hipGraphs (step + scout + plan; evaluator + gather), one stream sync per simulation.
    Same visit counts, bit for bit: the table returns what the evaluator returns for a position, and the evaluator's result for
    a row does not depend on the batch it sits in (tests/test_gpu_scouts.py)."""

    def __init__(self, engine: SelfPlayEngine, evaluator, version_fn=None, use_graph: bool = True, warmup: int = 3, device_loop=None):
        if not (getattr(evaluator, "returns_logits", False) and getattr(evaluator, "batched", False)):
            raise ValueError("scouts need a batched evaluator that returns logits (PolicyValueNet.evaluate_leaves_logits)")
        if getattr(engine, "n_scouts", 0) < 1:
            raise ValueError("the engine has no scout slots (SelfPlayEngine.set_scouts)")
        self.engine, self.evaluator, self.version_fn = engine, evaluator, version_fn
        self.use_graph = bool(use_graph and getattr(evaluator, "graph_safe", False) and engine.device.type == "cuda")
        self.warmup = warmup
        if version_fn is None and not hasattr(getattr(evaluator, "__self__", None), "weights_version") and not getattr(evaluator, "stateless", False):
            from .tools import log
            log("ScoutedSearch: the evaluator exposes no weights_version and no version_fn was given: evaluations cached for other weights "
                "cannot be told apart (pass version_fn=... if its weights can change)", "WARNING")
        self.version = self._weights_version()
        self._g_step = self._g_eval = self._g_run = self._g_eval_run = None
        self.evaluator_calls = self.simulations = 0
        # simulations that hit the table are repeated ON THE DEVICE (ccz_scouted_run: one launch per evaluator call instead of two
        # launches, a replay and a stream sync per simulation); CCZ_SCOUT_DEVICE_LOOP=0: the host loop (:meth:`simulate`), for A/B
        self.device_loop = (os.environ.get("CCZ_SCOUT_DEVICE_LOOP", "1") != "0") if device_loop is None else bool(device_loop)
        self.device_loop = self.device_loop and engine.B <= 16
        self._need = True

    def _weights_version(self):
        if self.version_fn is not None:
            return self.version_fn()
        return getattr(getattr(self.evaluator, "__self__", None), "weights_version", None)

    def _check_weights(self):
        v = self._weights_version()
        if v != self.version:          # cached evaluations (and captured addresses) of other weights must not reach the tree
            self.engine.clear_eval_cache()
            self.version = v
            self._g_eval = self._g_eval_run = None

    def _plan(self):
        self.engine.scout_and_plan()

    def _evaluate(self):
        e = self.engine
        logits, value = self.evaluator(e.leaf_input)
        e.gather_priors_planned(logits, value)

    def _capture(self, fn):
        dev = self.engine.device
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        return g

    def begin_move(self):
        """After a re-root / new position: select board 0's first leaf, scout, plan."""
        self._check_weights()
        self.engine.select_leaves()
        self._plan()
        if self.device_loop:
            self._need = self.engine.plan_state_of_board0() == 0

    def _warm_evaluator(self):
        e = self.engine
        dev = e.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(self.warmup):     # allocator / inference-copy warm-up outside the capture; results discarded
                self.evaluator(e.leaf_input)
        torch.cuda.current_stream(dev).wait_stream(side)

    def run(self, left: int, budget: int) -> int:
        """Up to ``budget`` simulations of the move (``left`` = how many it still has, the pending one included) with ONE launch
        sequence: the evaluator on all rows if board 0's pending leaf is not in the table, then ``ccz_scouted_run`` -- expand +
        backup, select, scout, probe, plan, again and again on the device until a leaf misses, the budget is used or the move's
        last simulation is backed up. Returns the number of simulations done (>= 1)."""
        e = self.engine
        e.set_run(min(budget, left), left)
        if self._need:
            if self.use_graph:
                if self._g_eval_run is None:
                    self._warm_evaluator()
                    self._g_eval_run = self._capture(lambda: (self._evaluate(), e.scouted_run_launch()))
                self._g_eval_run.replay()
            else:
                self._evaluate()
                e.scouted_run_launch()
            self.evaluator_calls += 1
        elif self.use_graph:
            if self._g_run is None:
                self._g_run = self._capture(e.scouted_run_launch)
            self._g_run.replay()
        else:
            e.scouted_run_launch()
        done, self._need = e.run_outcome()
        if done < 1:
            raise RuntimeError("ccz_scouted_run reported no simulation")
        self.simulations += done
        return done

    def simulate(self, last: bool):
        """One simulation of board 0: the evaluator only if its pending leaf is not in the table; then expand + backup (+ the next
        selection, scouting and plan unless ``last``)."""
        e = self.engine
        if e.plan_state_of_board0() == 0:
            if self.use_graph:
                if self._g_eval is None:
                    self._warm_evaluator()
                    self._g_eval = self._capture(self._evaluate)
                self._g_eval.replay()
            else:
                self._evaluate()
            self.evaluator_calls += 1
        self.simulations += 1
        if last:
            e.expand_backup_compact(None)
            return
        if self.use_graph:
            if self._g_step is None:
                self._g_step = self._capture(lambda: (e.step_compact(None), self._plan()))
            self._g_step.replay()
        else:
            e.step_compact(None)
            self._plan()


class BatchedSelfPlay:
    """``evaluator(leaf_input fp16 [B,17,7,10,9]) -> (prob f32 [B,2086], value f32 [B])`` on the device."""

    def __init__(self, evaluator, n_boards: int, n_playout: int = 400, c_puct: float = C_PUCT, eps: float = EPS,
                 alpha: float = ALPHA, temp: float = 1.0, seed: int = 0, board_id_base: int = 0, device: int = 0,
                 sampling: str = "device", use_graph: bool = False, version_fn=None, **engine_kw):
        """``version_fn() -> hashable``: what tells the evaluation cache (and a captured hipGraph) that the evaluator's weights
        changed; default: ``weights_version`` of the evaluator's owner. An evaluator that accepts a plan but exposes neither is
        refused: its cached evaluations could never be invalidated."""
        if sampling not in ("device", "numpy"):
            raise ValueError("sampling must be 'device' (Philox on the GPU) or 'numpy' (reference-exact host RNG)")
        self.evaluator = evaluator
        self.engine = SelfPlayEngine(n_boards, n_playout=n_playout, c_puct=c_puct, eps=eps, alpha=alpha, temp=temp,
                                     seed=seed, board_id_base=board_id_base, device=device, **engine_kw)
        self.B = n_boards
        self.n_playout = n_playout
        self.sampling = sampling
        self.eps, self.alpha, self.temp = eps, alpha, temp
        # reference-exact host sampling: one legacy RandomState per board (mcts.py:216-224 uses the global one)
        self.rngs = [np.random.RandomState((seed + board_id_base + b) % (2**32)) for b in range(n_boards)] if sampling == "numpy" else None
        self.use_graph = use_graph
        self._graph = None
        # planned evaluator boundary: the engine holds an evaluation cache (eval_cache_log2 > 0) and the evaluator can compute a
        # planned subset of the rows (PolicyValueNet.evaluate_leaves_logits): positions evaluated before are not sent through
        # the network again (include/cczero.h ccz_eval_plan). Same results, bit for bit.
        self.planned = bool(self.engine.eval_cache_log2 > 0 and getattr(evaluator, "accepts_plan", False)
                            and getattr(evaluator, "returns_logits", False))
        self.version_fn = version_fn
        if self.planned and version_fn is None and not hasattr(getattr(evaluator, "__self__", None), "weights_version") \
                and not getattr(evaluator, "stateless", False):
            raise ValueError("planned evaluator boundary (evaluation cache) with an evaluator that exposes no weights_version: "
                             "pass version_fn=... (what changes when its weights do), mark it `stateless = True`, or use eval_cache_log2=0")
        self._cache_version = self._evaluator_version()
        self._sim = 0        # simulations done of the current move
        self._leaf = None    # the pending leaf batch (None: the next simulation starts with select_leaves)
        self._acc = 0

    def _evaluator_version(self):
        """What tells the evaluation cache that the evaluator's weights changed: ``version_fn()`` if one was given, else the
        ``weights_version`` of the object the evaluator is a bound method of (``PolicyValueNet``)."""
        if self.version_fn is not None:
            return self.version_fn()
        return getattr(getattr(self.evaluator, "__self__", None), "weights_version", None)

    def _planned_eval(self, leaf):
        """(logits, value) of the planned rows; cached evaluations of other weights are dropped first."""
        e = self.engine
        v = self._evaluator_version()
        if v != self._cache_version:
            e.clear_eval_cache()
            self._cache_version = v
        return self.evaluator(leaf, plan=e.eval_plan())

    # one lockstep simulation of every board
    def simulate(self):
        e = self.engine
        leaf = e.select_leaves()
        if self.planned:
            e.expand_backup_planned(*self._planned_eval(leaf))
            return
        prob, value = self.evaluator(leaf)
        (e.expand_backup_logits if getattr(self.evaluator, "returns_logits", False) else e.expand_backup)(prob, value)

    def advance(self, steps: int, hooks=None, boundary=None, on_playout=None):
        """``steps`` lockstep simulations of every board from wherever the search stands in its move; a move that completes on the
        way (simulation ``n_playout``) is played: ``boundary()`` if given (it must call :meth:`finish_move` itself -- bench.py wraps
        the harvest / exchange and its timers around it), else :meth:`finish_move`. Returns the moves of the last move played in
        this call (device int32 [B]) or None.

        Launch sequence of a move: select, (evaluator, fused step) x (n-1), evaluator, expand_backup -- with an evaluation cache:
        (probe + plan, evaluator on the planned rows, softmax + gather + store, fused step).
        ``hooks(stage, sim)`` -- stage "eval0" / "eval1" / "step1" = before the evaluator, between evaluator and simulator
        kernel, after the simulator kernel of simulation index ``sim`` (0-based within its move): where bench.py records its HIP
        events and triggers the concurrent trainer. THE loop: ``run_move``, the bench, the soak and the tests all run this one."""
        e = self.engine
        n = self.n_playout
        logits = getattr(self.evaluator, "returns_logits", False)
        interval = max(1, n // 100)
        moves = None
        for _ in range(int(steps)):
            i = self._sim
            last = i + 1 == n
            if self._leaf is None:
                self._leaf = e.select_leaves()
                if self.use_graph and self._graph is None and not self.planned:
                    self._graph = GraphedStep(e, self.evaluator, version_fn=self.version_fn)
            if hooks is not None:
                hooks("eval0", i)
            if self._graph is not None and not last and hooks is None:
                self._graph.replay()   # evaluator + fused step as one captured graph (launch-bound small batches)
            else:
                if self.planned:   # cache probe + plan, the network on the planned rows only, softmax + gather + cache store
                    e.gather_priors_planned(*self._planned_eval(self._leaf))
                    value = None   # step / expand_backup then use the engine-owned leaf values (hits and fresh evaluations alike)
                else:
                    prob, value = self.evaluator(self._leaf)
                    if logits:     # the softmax + gather of the legal priors belongs to the evaluator side of the split
                        e.gather_priors(prob, value)
                if hooks is not None:
                    hooks("eval1", i)
                if last:
                    e.expand_backup_compact(value) if (logits or self.planned) else e.expand_backup(prob, value)
                    self._leaf = None
                else:
                    self._leaf = e.step_compact(value) if (logits or self.planned) else e.step(prob, value)
            if hooks is not None:
                hooks("step1", i)
            self._sim = 0 if last else i + 1
            self._acc += 1
            if on_playout is not None and (self._acc >= interval or last):
                try:
                    on_playout(self._acc)  # mcts.py:153-160
                except Exception:
                    pass
                self._acc = 0
            if last:
                self._acc = 0
                moves = boundary() if boundary is not None else self.finish_move()
        return moves

    def search(self, on_playout=None, hooks=None):
        """The simulations that are left of the current move (all ``n_playout`` from a move's start) WITHOUT playing it: the roots
        can be read (``engine.root_children()``) before :meth:`finish_move`."""
        left = self.n_playout - self._sim
        if left > 1:
            self.advance(left - 1, hooks=hooks, on_playout=on_playout)
        self.advance(1, hooks=hooks, on_playout=on_playout, boundary=lambda: None)

    def run_move(self, on_playout=None):
        """n_playout simulations then one move on every board. Returns the moves (device int32 [B])."""
        return self.advance(self.n_playout - self._sim, on_playout=on_playout)

    def watch(self, board_index: int, viewer):
        """Show ONE selected board of the batch in a viewer (``examples/viewer.py``'s ``ChessWindow`` or anything with
        ``update_board(svg, status)``): its position is pushed after every move (one small device read per move; nothing is
        read when no board is watched). The batched counterpart of ``Game.graphic`` (reference game.py:47-75)."""
        self._watch = (int(board_index), viewer)

    def _show(self, moves):
        b, viewer = self._watch
        from .boardsvg import board_svg
        from .tools import MOVE_FROM, MOVE_TO
        sq = self.engine.root_positions()[b]
        st = self.engine.game_status()
        mv = int(moves[b].item())
        lm = (int(MOVE_FROM[mv]), int(MOVE_TO[mv])) if mv >= 0 else None
        viewer.update_board(board_svg(sq, lm), f"board {b} - to move: {'red' if st['turn'][b] else 'black'} - ply: {int(st['plies'][b])}"
                            + (" - game over" if st["over"][b] else ""))

    def finish_move(self):
        self._sim, self._leaf, self._acc = 0, None, 0   # whatever leaf was pending belongs to the old root
        moves = self._finish_move()
        if getattr(self, "_watch", None) is not None:
            self._show(moves)
        return moves

    def _finish_move(self):
        e = self.engine
        if self.sampling == "device":
            return e.finish_move()
        # reference-exact: pi by the reference's own NumPy formula, Dirichlet + choice from legacy MT19937
        rc = e.root_children()
        st = e.game_status()
        forced = np.full(self.B, -1, np.int32)
        temps = np.ones(self.B, np.float64)
        for b in range(self.B):
            if st["over"][b]:
                continue
            k = int(rc["k"][b])
            temp = self.temp if (st["plies"][b] + 1) <= 30 else max(0.1, self.temp * 0.5)  # game.py:159
            temps[b] = temp
            visits = rc["visits"][b][:k].astype(np.int64)
            x = 1.0 / temp * np.log(visits + 1e-10)
            probs = np.exp(x - np.max(x))
            probs /= np.sum(probs)
            rs = self.rngs[b]
            p = (1 - self.eps) * probs + self.eps * rs.dirichlet(self.alpha * np.ones(k))
            forced[b] = int(rs.choice(rc["acts"][b][:k].astype(np.int64), p=p))
        return e.finish_move(forced_moves=forced, temps=temps)

    def harvest(self):
        return self.engine.harvest()

    def harvest_chunks(self, max_rows: int = 1 << 19):
        return self.engine.harvest_chunks(max_rows)

    def harvest_record_chunks(self, max_plies: int = 1 << 16):
        """Finished games as compact ply records (the exchange format, ``replay.RecordGatherer``)."""
        return self.engine.harvest_record_chunks(max_plies)
