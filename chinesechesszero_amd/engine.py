"""SelfPlayEngine -- thin Python handle on the gfx950 lockstep engine (libcczero.so).

B boards advance in lockstep; one simulation of every board is
``select_leaves() -> evaluator (PyTorch-ROCm, same stream) -> expand_backup()`` and one move is
``finish_move()``. PyTorch is only used for device memory and the stream.
Replaces, for all boards at once, reference mcts.py:101-178 + the cchess calls of SURVEY a17.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import MAX_LEGAL, NMOVES, SQ_STRIDE, CczError, Config, Stats, check


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        return C.c_void_p(t.data_ptr())
    if isinstance(t, np.ndarray):
        return C.c_void_p(t.ctypes.data)
    raise TypeError(type(t))


class SelfPlayEngine:
    def __init__(self, n_boards: int, n_playout: int = 400, c_puct: float = 5, eps: float = 0.25,
                 alpha: float = 0.2, temp: float = 1.0, seed: int = 0, board_id_base: int = 0,
                 device: int = 0, max_nodes: int = 0, max_depth: int = 0, max_plies: int = 0,
                 reference_quirks: bool = False, mirror: bool = True, reserve_nodes: int = 0,
                 move_rank="tools", plane_of_type="tools", value_f16: bool = False, type_rank="tools",
                 pawn_move_resets_clock="tools", perpetual_check="tools", eval_cache_log2: int = 0,
                 cache_verify: bool = False, strict: bool = False):
        """``move_rank`` (uint16[2086] permutation, None = ascending id) and ``plane_of_type`` (8 entries, None = type-1)
        are the run-time rule tables of ``ccz_config`` (ABI 2); the default "tools" takes the process-wide choice of
        :func:`chinesechesszero_amd.tools.set_rules`. ``eval_cache_log2`` = n > 0: an evaluation cache of 2^n positions
        (528 B each) for the planned evaluator boundary (:meth:`eval_plan`, ``include/cczero.h`` ccz_eval_plan);
        ``cache_verify``: its debug mode (CCZ_FLAG_CACHE_VERIFY): one hit in 128 is evaluated again and compared bit for bit
        (``stats()['cache_verified']`` / ``['cache_verify_mismatches']``). ``strict`` (CCZ_FLAG_STRICT): parity mode -- a kept
        subtree pruned to fit the node pool and a game adjudicated at ``max_plies`` (the reference knows neither: mcts.py:31-39,
        game.py:155) set sticky error bits that :meth:`check_healthy` raises on, instead of only moving ``pruned_subtrees`` /
        ``truncated_games``."""
        self.L = _lib.lib()
        if not torch.cuda.is_available():
            raise CczError("no GPU visible to PyTorch-ROCm; the engine has no CPU fallback")
        self.device = torch.device("cuda", device)
        self.B = int(n_boards)
        self.n_playout = int(n_playout)
        flags = (_lib.FLAG_REFERENCE_QUIRKS if reference_quirks else 0) | (0 if mirror else _lib.FLAG_NO_MIRROR) \
            | (_lib.FLAG_VALUE_F16 if value_f16 else 0) | (_lib.FLAG_CACHE_VERIFY if cache_verify else 0) | (_lib.FLAG_STRICT if strict else 0)
        self.strict = bool(strict)
        # value_f16: Q accumulated in float16 as on the reference's CUDA path
        self.mirror = mirror
        self.reference_quirks = bool(reference_quirks)
        self.max_plies = int(max_plies) if int(max_plies) > 0 else 2048  # recorded plies per game (ccz_config.max_plies)
        from . import tools
        if isinstance(move_rank, str):
            move_rank = tools.MOVE_RANK
        if isinstance(plane_of_type, str):
            plane_of_type = tools.PLANE_OF_TYPE
        if isinstance(type_rank, str):
            type_rank = tools.TYPE_RANK
        cfg = Config(n_boards=self.B, n_playout=self.n_playout, c_puct=float(c_puct), eps=float(eps),
                     alpha=float(alpha), temp=float(temp), max_nodes=int(max_nodes), max_depth=int(max_depth),
                     max_plies=int(max_plies), flags=flags, seed=int(seed) & (2**64 - 1),
                     board_id_base=int(board_id_base), device=int(device), reserve_nodes=int(reserve_nodes))
        self.move_rank = None if move_rank is None else np.ascontiguousarray(move_rank, dtype=np.uint16)
        if self.move_rank is not None:
            if self.move_rank.shape != (NMOVES,):
                raise ValueError("move_rank must have 2086 entries")
            cfg.move_rank_host = self.move_rank.ctypes.data
        self.plane_of_type = (0, 0, 1, 2, 3, 4, 5, 6) if plane_of_type is None else tuple(int(x) for x in plane_of_type)
        if plane_of_type is not None:
            cfg.plane_of_type = (C.c_uint8 * 8)(*self.plane_of_type)
        if isinstance(pawn_move_resets_clock, str):
            pawn_move_resets_clock = tools.PAWN_MOVE_RESETS_CLOCK
        self.pawn_move_resets_clock = bool(pawn_move_resets_clock)
        if isinstance(perpetual_check, str):
            perpetual_check = tools.PERPETUAL_CHECK
        self.perpetual_check = bool(perpetual_check)
        cfg.rule_flags = (_lib.RULE_PAWN_MOVE_RESETS_CLOCK if self.pawn_move_resets_clock else 0) \
            | (_lib.RULE_PERPETUAL_CHECK if self.perpetual_check else 0)
        self.type_rank = None if type_rank is None else tuple(int(x) for x in type_rank)
        if self.type_rank is not None:
            if len(self.type_rank) != 8:
                raise ValueError("type_rank must have 8 entries")
            cfg.type_rank = (C.c_uint8 * 8)(*self.type_rank)
        self.eval_cache_log2 = int(eval_cache_log2)
        cfg.eval_cache_log2 = self.eval_cache_log2
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.L.ccz_create(C.byref(cfg), C.byref(h)))
        self.h = h
        # the evaluation plan of the current step (planned evaluator boundary): rows the evaluator has to compute, their number
        self.miss_rows = torch.zeros((self.B,), dtype=torch.int32, device=self.device)
        self.n_miss = torch.zeros((1,), dtype=torch.int32, device=self.device)
        # torch-owned boundary buffers (evaluator input / outputs, move buffers)
        self.leaf_input = torch.zeros((self.B, 17, 7, 10, 9), dtype=torch.float16, device=self.device)
        self.moves_out = torch.full((self.B,), -1, dtype=torch.int32, device=self.device)
        self._forced = torch.full((self.B,), -1, dtype=torch.int32, device=self.device)
        self._temps = torch.ones((self.B,), dtype=torch.float64, device=self.device)

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.L.ccz_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ games
    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        check(self.L.ccz_reset(self.h, self._stream(), _ptr(m)))

    def reset_tree(self, mask=None):
        """Fresh root on the masked boards (all if None); position, history and game record stay (mcts.py:176-178)."""
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        check(self.L.ccz_reset_tree(self.h, self._stream(), _ptr(m)))

    def set_position(self, board: int, squares, turn: int, halfmove: int = 0):
        sq = np.ascontiguousarray(squares, dtype=np.uint8)
        assert sq.shape == (90,)
        check(self.L.ccz_set_position(self.h, self._stream(), int(board), _ptr(sq), int(turn), int(halfmove)))

    # ------------------------------------------------------------------ one simulation
    def select_leaves(self) -> torch.Tensor:
        """PUCT descent + leaf rules + evaluator input for all boards; returns [B,17,7,10,9] fp16."""
        check(self.L.ccz_select_leaves(self.h, self._stream(), _ptr(self.leaf_input)))
        return self.leaf_input

    def expand_backup(self, prob: torch.Tensor, value: torch.Tensor):
        """prob float32 [B,2086] (= exp(log_act_probs)), value float32 [B]."""
        if prob.dtype != torch.float32 or value.dtype != torch.float32:
            raise TypeError("prob and value must be float32")
        if tuple(prob.shape) != (self.B, NMOVES) or value.numel() != self.B:
            raise ValueError(f"prob must be [{self.B},{NMOVES}] and value [{self.B}]")
        if not (prob.is_cuda and value.is_cuda and prob.is_contiguous() and value.is_contiguous()):
            raise ValueError("prob/value must be contiguous device tensors")
        check(self.L.ccz_expand_backup(self.h, self._stream(), _ptr(prob), _ptr(value)))

    def step(self, prob: torch.Tensor, value: torch.Tensor) -> torch.Tensor:
        """Fused expand_backup(prob, value) for the pending leaf + select_leaves() for the next simulation."""
        if prob.dtype != torch.float32 or value.dtype != torch.float32:
            raise TypeError("prob and value must be float32")
        if tuple(prob.shape) != (self.B, NMOVES) or value.numel() != self.B:
            raise ValueError(f"prob must be [{self.B},{NMOVES}] and value [{self.B}]")
        if not (prob.is_cuda and value.is_cuda and prob.is_contiguous() and value.is_contiguous()):
            raise ValueError("prob/value must be contiguous device tensors")
        check(self.L.ccz_step(self.h, self._stream(), _ptr(prob), _ptr(value), _ptr(self.leaf_input)))
        return self.leaf_input

    def _check_logits(self, logits, value):
        if logits.dtype not in (torch.float16, torch.float32) or value.dtype != torch.float32:
            raise TypeError("logits must be float16/float32 and value float32")
        if tuple(logits.shape) != (self.B, NMOVES) or value.numel() != self.B:
            raise ValueError(f"logits must be [{self.B},{NMOVES}] and value [{self.B}]")
        if not (logits.is_cuda and value.is_cuda and logits.is_contiguous() and value.is_contiguous()):
            raise ValueError("logits/value must be contiguous device tensors")
        return 1 if logits.dtype == torch.float16 else 0

    def step_logits(self, logits: torch.Tensor, value: torch.Tensor) -> torch.Tensor:
        """Compact boundary: policy-head LOGITS in; the engine takes exp(log_softmax) of the legal ids itself."""
        self.gather_priors(logits, value)
        return self.step_compact(value)

    def expand_backup_logits(self, logits: torch.Tensor, value: torch.Tensor):
        self.gather_priors(logits, value)
        check(self.L.ccz_expand_backup_compact(self.h, self._stream(), _ptr(value)))

    def gather_priors(self, logits: torch.Tensor, value: torch.Tensor):
        """exp(log_softmax(logits)) of the pending leaf's legal ids -> the engine's [B,128] prior row."""
        f16 = self._check_logits(logits, value)
        check(self.L.ccz_gather_priors(self.h, self._stream(), _ptr(logits), f16))

    def expand_backup_compact(self, value: torch.Tensor | None):
        """``value`` None: the engine-owned leaf values of the planned boundary."""
        check(self.L.ccz_expand_backup_compact(self.h, self._stream(), _ptr(value)))

    def step_compact(self, value: torch.Tensor | None) -> torch.Tensor:
        """``value`` None: the engine-owned leaf values of the planned boundary (:meth:`gather_priors_planned`)."""
        check(self.L.ccz_step_compact(self.h, self._stream(), _ptr(value), _ptr(self.leaf_input)))
        return self.leaf_input

    # ------------------------------------------------------------------ planned evaluator boundary (evaluation cache)
    def eval_plan(self):
        """Probe the evaluation cache for the pending leaves and plan the evaluator's batch: returns the device tensors
        ``(miss_rows int32 [B], n_miss int32 [1])`` -- the boards whose leaves still need the network (deduplicated, ascending)
        and their number. No host sync: the evaluator's kernels read the count on the device."""
        check(self.L.ccz_eval_plan(self.h, self._stream(), _ptr(self.miss_rows), _ptr(self.n_miss)))
        return self.miss_rows, self.n_miss

    # ------------------------------------------------------------------ scouts (one game at a time: include/cczero.h ccz_scout)
    def set_scouts(self, n_scouts: int):
        """The last ``n_scouts`` boards become scout slots (no tree, no game): the simulator entry points then run on the first
        ``B - n_scouts`` boards only. Needs an evaluation cache."""
        check(self.L.ccz_set_scouts(self.h, int(n_scouts)))
        self.n_scouts = int(n_scouts)
        if not hasattr(self, "_plan_state_host"):
            self._plan_state_host = torch.zeros((self.B,), dtype=torch.int32).pin_memory()   # written by k_cache_plan_scouted

    def scout(self):
        """Hand every scout slot its pending leaf: the next unvisited sibling(s) of the pending leaf of the board it scouts for --
        what that board's next simulations through the same parent will ask for (mcts.py:47-48: first maximum = insertion order)."""
        check(self.L.ccz_scout(self.h, self._stream(), _ptr(self.leaf_input)))

    def plan_scouted_launch(self):
        """Probe the table for every slot and plan the step's evaluator call (``ccz_eval_plan_scouted``; launches only). The plan
        state of the searched boards is written straight into pinned HOST memory by the plan kernel (no copy node: capturable)."""
        check(self.L.ccz_eval_plan_scouted(self.h, self._stream(), _ptr(self.miss_rows), _ptr(self.n_miss),
                                           C.c_void_p(self._plan_state_host.data_ptr())))

    def scout_and_plan(self):
        """:meth:`scout` + :meth:`plan_scouted_launch` as one launch (``ccz_scout_and_plan``)."""
        check(self.L.ccz_scout_and_plan(self.h, self._stream(), _ptr(self.leaf_input), _ptr(self.miss_rows), _ptr(self.n_miss),
                                        C.c_void_p(self._plan_state_host.data_ptr())))

    def scouted_run_launch(self):
        """Simulations in ONE launch (``ccz_scouted_run``): step + scout + probe + plan repeated on the device while board 0's next
        leaf is in the table. How many at most, and how many the move has left, go through pinned host memory -- the launch is
        capturable; write them with :meth:`set_run` before the launch or a replay, read the outcome with :meth:`run_outcome`."""
        if not hasattr(self, "_run_host"):
            self.set_run(1, 1)
        check(self.L.ccz_scouted_run(self.h, self._stream(), _ptr(self.leaf_input), _ptr(self.miss_rows), _ptr(self.n_miss),
                                     C.c_void_p(self._plan_state_host.data_ptr()), C.c_void_p(self._run_host.data_ptr())))

    def set_run(self, budget: int, left: int):
        """``budget``: simulations the next scouted run may do at most; ``left``: simulations left in this move, the pending one included."""
        if budget < 1 or left < 1:
            raise ValueError("scouted run: budget and simulations left must be >= 1")
        if not hasattr(self, "_run_host"):
            self._run_host = torch.zeros((4,), dtype=torch.int32).pin_memory()
            self._run_np = self._run_host.numpy()
        self._run_np[0] = budget
        self._run_np[1] = left

    def run_outcome(self):
        """Wait for the stream; ``(simulations done by the last scouted run, whether board 0's leaf needs the evaluator now)``."""
        torch.cuda.current_stream(self.device).synchronize()
        return int(self._run_np[2]), bool(self._run_np[3])

    def plan_state_of_board0(self) -> int:
        """Wait for the stream and read board 0's plan state: 0 = its leaf needs the evaluator -- run it on ALL ``B`` rows of
        ``leaf_input`` and hand the result to :meth:`gather_priors_planned` --, 1 = table hit, 2 = no evaluation needed."""
        torch.cuda.current_stream(self.device).synchronize()
        return int(self._plan_state_host[0])

    def plan_states(self) -> np.ndarray:
        """Wait for the stream; the plan states of ALL searched boards (0 miss / 1 hit / 2 nothing to evaluate): the evaluator has to
        run iff any of them is 0 (the host loop of the package searches one board and reads :meth:`plan_state_of_board0`)."""
        torch.cuda.current_stream(self.device).synchronize()
        return self._plan_state_host[: self.B - self.n_scouts].numpy().copy()

    def gather_priors_planned(self, logits: torch.Tensor, value: torch.Tensor):
        """``logits`` [B,2086] / ``value`` [B] as the planned evaluator returns them: COMPACT, row i = board miss_rows[i]."""
        f16 = self._check_logits(logits, value)
        check(self.L.ccz_gather_priors_planned(self.h, self._stream(), _ptr(logits), f16, _ptr(value)))

    def step_planned(self, logits: torch.Tensor, value: torch.Tensor) -> torch.Tensor:
        self.gather_priors_planned(logits, value)
        return self.step_compact(None)

    def expand_backup_planned(self, logits: torch.Tensor, value: torch.Tensor):
        self.gather_priors_planned(logits, value)
        check(self.L.ccz_expand_backup_compact(self.h, self._stream(), None))

    def clear_eval_cache(self):
        """Forget every cached evaluation (the evaluator's weights changed)."""
        check(self.L.ccz_eval_cache_clear(self.h, self._stream()))

    # ------------------------------------------------------------------ once per move
    def finish_move(self, forced_moves=None, temps=None, keep_tree: bool = True) -> torch.Tensor:
        """Record pi, choose (or accept) the move, re-root, push, detect game end. Returns moves int32[B] (device)."""
        f = t = None
        if forced_moves is not None:
            self._forced.copy_(torch.as_tensor(np.asarray(forced_moves, dtype=np.int32)) if not isinstance(forced_moves, torch.Tensor) else forced_moves)
            f = self._forced
        if temps is not None:
            self._temps.copy_(torch.as_tensor(np.asarray(temps, dtype=np.float64)) if not isinstance(temps, torch.Tensor) else temps)
            t = self._temps
        check(self.L.ccz_finish_move(self.h, self._stream(), _ptr(f), _ptr(t), _ptr(self.moves_out), 1 if keep_tree else 0))
        return self.moves_out

    # ------------------------------------------------------------------ inspection (sync)
    def root_children(self):
        B = self.B
        k = np.zeros(B, np.int32)
        acts = np.zeros((B, MAX_LEGAL), np.uint16)
        visits = np.zeros((B, MAX_LEGAL), np.int32)
        q = np.zeros((B, MAX_LEGAL), np.float32)
        p = np.zeros((B, MAX_LEGAL), np.float32)
        rn = np.zeros(B, np.int32)
        check(self.L.ccz_root_children(self.h, self._stream(), _ptr(k), _ptr(acts), _ptr(visits), _ptr(q), _ptr(p), _ptr(rn)))
        return {"k": k, "acts": acts, "visits": visits, "q": q, "prior": p, "root_visits": rn}

    def root_pi(self, temps=None) -> np.ndarray:
        pi = np.zeros((self.B, MAX_LEGAL), np.float64)
        t = None if temps is None else np.ascontiguousarray(np.broadcast_to(np.asarray(temps, np.float64), (self.B,)))
        check(self.L.ccz_root_pi(self.h, self._stream(), _ptr(t), _ptr(pi)))
        return pi

    def game_status(self):
        B = self.B
        over = np.zeros(B, np.uint8)
        winner = np.zeros(B, np.int8)
        plies = np.zeros(B, np.int32)
        turn = np.zeros(B, np.uint8)
        check(self.L.ccz_game_status(self.h, self._stream(), _ptr(over), _ptr(winner), _ptr(plies), _ptr(turn)))
        return {"over": over, "winner": winner, "plies": plies, "turn": turn}

    def root_positions(self) -> np.ndarray:
        sq = np.zeros((self.B, SQ_STRIDE), np.uint8)
        check(self.L.ccz_root_positions(self.h, self._stream(), _ptr(sq)))
        return sq[:, :90].copy()

    def leaf_info(self):
        B = self.B
        status = np.zeros(B, np.uint8)
        k = np.zeros(B, np.int32)
        ids = np.zeros((B, MAX_LEGAL), np.uint16)
        depth = np.zeros(B, np.int32)
        check(self.L.ccz_leaf_info(self.h, self._stream(), _ptr(status), _ptr(k), _ptr(ids), _ptr(depth)))
        return {"status": status, "k": k, "ids": ids, "depth": depth}

    def leaf_priors(self, values: bool = True):
        """What the compact / planned boundary hands the tree for the pending leaves, after ``gather_priors[_planned]``:
        ``(prior float32 [B,128] aligned with leaf_info()['ids'], value float32 [B] or None)``; ``values`` needs an evaluation
        cache (engine-owned leaf values). Syncs (tests)."""
        pri = np.zeros((self.B, MAX_LEGAL), np.float32)
        val = np.zeros(self.B, np.float32) if values else None
        check(self.L.ccz_leaf_priors(self.h, self._stream(), _ptr(pri), _ptr(val)))
        return pri, val

    def leaf_keys(self):
        """(keys int64 [B], status uint8 [B]) of the pending leaves as device tensors (no sync): equal keys = equal evaluator
        input (position + side to move)."""
        keys = torch.empty((self.B,), dtype=torch.int64, device=self.device)
        status = torch.empty((self.B,), dtype=torch.uint8, device=self.device)
        check(self.L.ccz_leaf_keys(self.h, self._stream(), _ptr(keys), _ptr(status)))
        return keys, status

    def stats(self) -> dict:
        s = Stats()
        check(self.L.ccz_get_stats(self.h, self._stream(), C.byref(s)))
        d = {f: getattr(s, f) for f, _ in Stats._fields_ if f != "reserved"}
        if s.reserved:
            d["bounds_line"] = int(s.reserved)  # bounds-checked diagnostic build: where cczero_kernels.h indexed out of range
        return d

    def check_healthy(self):
        e = self.stats()["error_flags"]
        if e:
            msgs = [m for bit, m in _lib.ERR_BITS.items() if e & bit]
            raise CczError(f"engine error flags {e}: " + "; ".join(msgs) + (f" (cczero_kernels.h:{self.stats()['bounds_line']})" if e & 128 else ""))

    # ------------------------------------------------------------------ training tuples
    def harvest_chunks(self, max_rows: int = 1 << 19):
        """Yield (states fp16 [R,17,7,10,9], pi f32 [R,2086], z f32 [R]) chunks of at most ``max_rows`` rows until no
        finished game is left; harvested boards restart. (2^19 rows = 15.6 GB: many boards can reach the ply cap in
        the same move, so the rows of one harvest are bounded by the buffer, not by the number of finished games.)"""
        while True:
            rows = C.c_int64(0)
            check(self.L.ccz_harvest_rows(self.h, self._stream(), C.byref(rows)))
            total = int(rows.value)
            if total == 0:
                return
            cap = min(total, int(max_rows))
            while True:
                states = torch.empty((cap, 17, 7, 10, 9), dtype=torch.float16, device=self.device)
                pi = torch.empty((cap, NMOVES), dtype=torch.float32, device=self.device)
                z = torch.empty((cap,), dtype=torch.float32, device=self.device)
                got = C.c_int64(0)
                rc = self.L.ccz_harvest(self.h, self._stream(), _ptr(states), _ptr(pi), _ptr(z), cap, C.byref(got))
                if rc == -5 and cap < total:  # one single game is longer than the chunk: grow to fit it
                    cap = min(total, cap * 2)
                    continue
                check(rc)
                break
            R = int(got.value)
            yield states[:R], pi[:R], z[:R]

    def harvest(self, max_rows: int = 1 << 19):
        """Tuples of all finished games -> (states fp16 [R,17,7,10,9], pi f32 [R,2086], z f32 [R]); restarts those
        boards. Raises if more than ``max_rows`` rows are pending: iterate :meth:`harvest_chunks` instead."""
        rows = C.c_int64(0)
        check(self.L.ccz_harvest_rows(self.h, self._stream(), C.byref(rows)))
        if rows.value > max_rows:
            raise CczError(f"{rows.value} rows pending (> {max_rows}): use harvest_chunks() to bound device memory")
        chunks = list(self.harvest_chunks(max_rows))
        if not chunks:
            return (torch.empty((0, 17, 7, 10, 9), dtype=torch.float16, device=self.device),
                    torch.empty((0, NMOVES), dtype=torch.float32, device=self.device),
                    torch.empty((0,), dtype=torch.float32, device=self.device))
        if len(chunks) == 1:
            return chunks[0]
        return tuple(torch.cat([c[i] for c in chunks]) for i in range(3))


    def harvest_record_chunks(self, max_plies: int = 1 << 16):
        """Yield uint8 [P, 880] tensors of compact ply records (``include/cczero.h`` CCZ_REC_*: whole finished games, plies in
        order) of at most ``max_plies`` records until no finished game is left; harvested boards restart. The wire format of
        the multi-GPU exchange: :func:`expand_records` rebuilds from them, byte for byte, the rows :meth:`harvest_chunks` yields."""
        mul = 2 if self.mirror else 1
        while True:
            rows = C.c_int64(0)
            check(self.L.ccz_harvest_rows(self.h, self._stream(), C.byref(rows)))
            total = int(rows.value) // mul
            if total == 0:
                return
            cap = min(total, int(max_plies))
            while True:
                rec = torch.empty((cap, _lib.REC_BYTES), dtype=torch.uint8, device=self.device)
                got = C.c_int64(0)
                rc = self.L.ccz_harvest_records(self.h, self._stream(), _ptr(rec), cap, C.byref(got))
                if rc == -5 and cap < total:  # one single game is longer than the chunk: grow to fit it
                    cap = min(total, cap * 2)
                    continue
                check(rc)
                break
            yield rec[:int(got.value)]

    def record_flags(self) -> int:
        """The ``flags`` :func:`expand_records` needs to reproduce this engine's :meth:`harvest` (quirk mode, mirror)."""
        return (_lib.FLAG_REFERENCE_QUIRKS if self.reference_quirks else 0) | (0 if self.mirror else _lib.FLAG_NO_MIRROR)


def rows_of_records(n_plies: int, flags: int = 0) -> int:
    return int(n_plies) * (1 if flags & _lib.FLAG_NO_MIRROR else 2)


def game_aligned_chunks(records: torch.Tensor, max_plies: int):
    """Split records uint8 [P, 880] (whole games) into views of at most ``max_plies`` records that hold whole games each
    (a single longer game becomes its own chunk). Reads one 2-byte header word per cut."""
    P, lo = int(records.shape[0]), 0
    while lo < P:
        hi = min(P, lo + int(max_plies))
        if hi < P:
            t = int(records[hi, _lib.REC_HDR:_lib.REC_HDR + 2].cpu().view(torch.int16).item()) & 0xffff  # ply index of record hi
            if hi - t > lo:
                hi -= t       # cut in front of the game that holds record hi
            else:             # the game starting at lo is longer than max_plies: take it whole (T at header bytes 2..3)
                T = int(records[lo, _lib.REC_HDR + 2:_lib.REC_HDR + 4].cpu().view(torch.int16).item()) & 0xffff
                hi = min(P, lo + max(T, 1))
        yield records[lo:hi]
        lo = hi


def expand_records(records: torch.Tensor, flags: int = 0, plane_of_type=None, out=None, head_row: int = 0, bad=None):
    """Compact ply records (uint8 [P, 880] on the GPU, whole games) -> the dense training rows ``ccz_harvest`` would have
    written for those games: (states fp16 [R,17,7,10,9], pi f32 [R,2086], z f32 [R]), R = P x (1 or 2 with mirror images).
    Stateless (no engine: the records may come from another rank). ``out=(states, pi, z)`` writes into existing arrays as a
    ring: row i goes to (head_row + i) % len(z). ``bad``: int32 device tensor [1] counting records of cut games (skipped).
    Asynchronous on the current stream. reference game.py:213-237 + collect.py:64-131 (preprocess, flip_data)."""
    L = _lib.lib()
    if not (records.is_cuda and records.dtype == torch.uint8 and records.is_contiguous()):
        raise ValueError("records must be a contiguous uint8 device tensor")
    if records.numel() % _lib.REC_BYTES:
        raise ValueError("records must hold whole 880-byte ply records")
    P = records.numel() // _lib.REC_BYTES
    R = rows_of_records(P, flags)
    dev = records.device
    if out is None:
        states = torch.empty((R, 17, 7, 10, 9), dtype=torch.float16, device=dev)
        pi = torch.empty((R, NMOVES), dtype=torch.float32, device=dev)
        z = torch.empty((R,), dtype=torch.float32, device=dev)
        ring = 0
    else:
        states, pi, z = out
        ring = int(z.shape[0])
        if not (states.is_contiguous() and pi.is_contiguous() and z.is_contiguous() and states.shape[0] == ring and pi.shape[0] == ring
                and states.dtype == torch.float16 and pi.dtype == torch.float32 and z.dtype == torch.float32):
            raise ValueError("out must be contiguous (states fp16 [N,17,7,10,9], pi f32 [N,2086], z f32 [N])")
    pot = None if plane_of_type is None else (C.c_uint8 * 8)(*[int(x) for x in plane_of_type])
    with torch.cuda.device(dev):
        check(L.ccz_expand_records(C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), _ptr(records), P, int(flags), pot,
                                   _ptr(states), _ptr(pi), _ptr(z), ring, int(head_row) if ring else 0, _ptr(bad)))
    return states, pi, z


# ---------------------------------------------------------------------- stateless batch rules
def legal_moves(squares, turn, halfmove=None, device: int = 0):
    """Legal-move bitmask / count / flags of n positions on the GPU (replaces cchess legal_moves).

    squares uint8 [n,90], turn [n]. Returns (mask bool [n,2086], count int32 [n], flags uint8 [n]).
    """
    L = _lib.lib()
    if not torch.cuda.is_available():
        raise CczError("no GPU visible to PyTorch-ROCm; the engine has no CPU fallback")
    dev = torch.device("cuda", device)
    sq = np.zeros((len(squares), SQ_STRIDE), np.uint8)
    sq[:, :90] = np.asarray(squares, np.uint8)
    n = sq.shape[0]
    d_sq = torch.from_numpy(sq).to(dev)
    d_turn = torch.from_numpy(np.ascontiguousarray(turn, dtype=np.uint8)).to(dev)
    d_half = None if halfmove is None else torch.from_numpy(np.ascontiguousarray(halfmove, dtype=np.int32)).to(dev)
    d_mask = torch.zeros((n, _lib.MASK_WORDS), dtype=torch.int32, device=dev)
    d_cnt = torch.zeros((n,), dtype=torch.int32, device=dev)
    d_flags = torch.zeros((n,), dtype=torch.uint8, device=dev)
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    check(L.ccz_legal_moves(s, n, _ptr(d_sq), _ptr(d_turn), _ptr(d_half), _ptr(d_mask), _ptr(d_cnt), _ptr(d_flags)))
    words = d_mask.cpu().numpy().view(np.uint32)
    bits = np.unpackbits(words.view(np.uint8), axis=1, bitorder="little")[:, :NMOVES].astype(bool)
    return bits, d_cnt.cpu().numpy(), d_flags.cpu().numpy()


def apply_moves(squares, turn, move_ids, device: int = 0):
    L = _lib.lib()
    dev = torch.device("cuda", device)
    sq = np.zeros((len(squares), SQ_STRIDE), np.uint8)
    sq[:, :90] = np.asarray(squares, np.uint8)
    n = sq.shape[0]
    d_sq = torch.from_numpy(sq).to(dev)
    d_turn = torch.from_numpy(np.ascontiguousarray(turn, dtype=np.uint8)).to(dev)
    d_ids = torch.from_numpy(np.ascontiguousarray(move_ids, dtype=np.int32)).to(dev)
    d_cap = torch.zeros((n,), dtype=torch.uint8, device=dev)
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    check(L.ccz_apply_moves(s, n, _ptr(d_sq), _ptr(d_turn), _ptr(d_ids), _ptr(d_cap)))
    return d_sq.cpu().numpy()[:, :90], d_turn.cpu().numpy(), d_cap.cpu().numpy()
