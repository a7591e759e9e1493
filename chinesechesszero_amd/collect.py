"""Host mirror of reference collect.py: ``CollectPipeline`` (collect.py:26-198).

Two modes behind the same class:
  * ``n_boards == 1``: the reference's own control flow -- ``Game.start_self_play`` with an ``MCTS_AI``
    player, ``preprocess`` and ``flip_data`` on the host -- for drop-in use and parity tests;
  * ``n_boards > 1``: the MI355X-native path -- ``BatchedSelfPlay`` runs all games in lockstep on the
    GPU, ``harvest`` materialises (state, pi, z) rows incl. the mirror images on the device; with several ranks
    (``python -m torch.distributed.run --nproc-per-node N -m chinesechesszero_amd.collect --boards 4096``: the N collectors of the
    reference's README.md:31-48 as one job) finished games are all-gathered over RCCL as compact records without any rank waiting
    for another (``replay.AsyncRecordExchange``) and rank 0 stores the union: what N collectors appending to one file produce.
Rows go to a ``TupleSink``: shard files while collecting, merged by ``TupleSink.finalize()`` into ``states.npy /
mcts.npy / winners.npy / meta.json`` with the array names, dtypes and ``meta.json`` keys of the reference's
convert.py:84-99 (what dataset.py:45-89 reads). h5py is not part of this image: the per-game HDF5 groups of
collect.py:146-167 are not written.
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from .game import RED, Board, Game
from .mcts import MCTS_AI
from .net import PolicyValueNet
from .parameters import C_PUCT, DATA_DIR, MODEL_DIR, PLAYOUT
from .tools import flip_map, log


class TupleSink:
    """Append-only store of training rows in the trainer's on-disk format.

    ``append`` streams every batch of rows to its own shard file (host memory stays bounded by one batch: a
    4096-board run produces tens of GB of rows) and is all a collector does while it runs -- the reference likewise
    appends one group per game (collect.py:146-167) and leaves the flattening to its offline converter.
    ``finalize`` is that converter step (convert.py:21-107): old files + shards are merged through memory maps into
    ``states.npy / mcts.npy / winners.npy`` and ``meta.json`` is written LAST with the keys of convert.py:89-97
    (``total_count, states_shape, states_dtype, mcts_shape, mcts_dtype, winners_shape, winners_dtype``; plus
    ``iters``, the game counter the reference keeps in ``data.h5`` attrs). ``mcts.npy`` is float64 by default, the
    dtype the reference stores (mcts.py:212 ``np.zeros(2086)`` kept as is by collect.py:157-160); ``pi_dtype=np.float32``
    halves the file. One collector per directory (a pid lock file; a dead collector's lock is taken over and its shards
    adopted). ``finalize`` is journaled: validated before the first write, idempotent after a crash at any point.

    ``append_records`` (round 5) is what the batched collector writes while it runs: the finished games as COMPACT ply records
    (880 B per ply, ``include/cczero.h`` CCZ_REC_*) instead of the dense rows they stand for (2 x 29,768 B + a float64 pi per
    ply): 12.5 MB per move of 4096 boards instead of 1.07 GB -- which took the build container's host 10.4 s to convert and
    write, longer than the move's search on the GPU. ``finalize`` expands the record shards with the GPU expander
    (``ccz_expand_records``: byte for byte the rows ``ccz_harvest`` writes, tests/test_gpu_harvest.py) before it merges; the
    reference, too, converts offline (convert.py).
    """

    ARRAYS = {"states": ("_s.npy", np.float16, (17, 7, 10, 9)), "mcts": ("_p.npy", None, (2086,)), "winners": ("_z.npy", np.float32, ())}

    def __init__(self, out_dir: str = DATA_DIR, pi_dtype=np.float64):
        self.out_dir = out_dir
        self.pi_dtype = np.dtype(pi_dtype)
        self._shards: list[tuple[str, int]] = []
        self._rshards: list[tuple[str, int, int, tuple]] = []   # record shards: (path, plies, flags, plane_of_type)
        self._next = 0
        self.games = 0
        os.makedirs(out_dir, exist_ok=True)
        self._take_lock()
        state = os.path.join(out_dir, "collect_state.json")
        if os.path.exists(state):
            with open(state, encoding="utf-8") as f:
                self.games = int(json.load(f).get("iters", 0))
        self._recover()  # a merge that a crash interrupted is completed from its journal before anything else is looked at
        meta = os.path.join(out_dir, "meta.json")
        m = {}
        if os.path.exists(meta):
            with open(meta, encoding="utf-8") as f:
                m = json.load(f)
            self.games = max(self.games, int(m.get("iters", 0)))
        self._merged_rows = self._rows_on_disk(m.get("total_count"))
        # shards a previous (crashed or still unmerged) collector left behind are part of the data set; a LIVE collector's
        # directory is refused by the lock above
        for name in sorted(os.listdir(out_dir)):
            if name.startswith(".rshard_") and name.endswith(".npy"):
                got = self._parse_rshard(os.path.join(out_dir, name))
                if got is not None:
                    self._rshards.append(got)
        # Shard names never repeat within a directory: the sequence number starts above every one already used there, by ANY process id
        # and in ANY of the three name forms. (a restarted collector with the same pid -- usual in containers -- started at
        # 0 and only looked at `.rshard_` names; the dense shards `.shard_r<pid>_<seq>_..` of an expansion whose record shard was already
        # gone were adopted below, a new record shard then took their tag, and the next finalize deleted them as "leftovers of an
        # interrupted expansion of THIS shard": finished games lost.)
        self._next = self._first_free_seq(os.listdir(out_dir))
        unexpanded = tuple(".shard_r" + os.path.basename(p)[len(".rshard_"):-len(".npy")] + "_" for p, _, _, _ in self._rshards)
        for name in sorted(os.listdir(out_dir)):
            if name.startswith(".shard_") and name.endswith("_z.npy"):
                base = os.path.join(out_dir, name[:-len("_z.npy")])
                if unexpanded and name.startswith(unexpanded):
                    continue   # dense rows of a record shard whose expansion was interrupted: it is still there and will be expanded again
                if all(os.path.exists(base + sfx) for sfx, _, _ in self.ARRAYS.values()):
                    self._shards.append((base, int(np.load(base + "_z.npy", mmap_mode="r").shape[0])))

    @staticmethod
    def _first_free_seq(names) -> int:
        """1 + the highest sequence number in the shard names ``.rshard_<pid>_<seq>_..``, ``.shard_r<pid>_<seq>_..`` and
        ``.shard_<pid>_<seq>_..`` among ``names`` (0 for a directory without shards)."""
        import re
        pat = re.compile(r"^\.(?:rshard_|shard_r|shard_)(\d+)_(\d+)[_.]")
        top = -1
        for name in names:
            m = pat.match(name)
            if m:
                top = max(top, int(m.group(2)))
        return top + 1

    # ---- one collector per directory ------------------------------------------------------------------
    def _take_lock(self):
        path = os.path.join(self.out_dir, ".collector.lock")
        if os.path.exists(path):
            try:
                with open(path, encoding="utf-8") as f:
                    pid = int(f.read().strip() or 0)
            except (OSError, ValueError):
                pid = 0
            alive = False
            if pid and pid != os.getpid():
                try:
                    os.kill(pid, 0)
                    alive = True
                except ProcessLookupError:
                    alive = False
                except PermissionError:
                    alive = True
            if alive:
                raise RuntimeError(f"{self.out_dir} is in use by collector process {pid} (its shards would be adopted and deleted); "
                                   "use another data_dir or stop that collector")
        with open(path, "w", encoding="utf-8") as f:
            f.write(str(os.getpid()))
        self._lock = path

    def close(self):
        """Release the directory (shards that were not finalized stay on disk and are adopted by the next sink)."""
        path = getattr(self, "_lock", None)
        if path and os.path.exists(path):
            try:
                with open(path, encoding="utf-8") as f:
                    mine = f.read().strip() == str(os.getpid())
                if mine:
                    os.remove(path)
            except OSError:
                pass
        self._lock = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- what is on disk ------------------------------------------------------------------------------
    def _journal(self) -> str:
        return os.path.join(self.out_dir, "merge_journal.json")

    def _paths(self) -> dict:
        return {k: os.path.join(self.out_dir, k + ".npy") for k in self.ARRAYS}

    def _rows_on_disk(self, meta_total=None) -> int:
        """Rows of the merged arrays, taken from the FILES (all three must agree); ``meta_total`` (meta.json's
        ``total_count``, if it has one) must agree with them too. A directory written by an older layout (no ``total_count``,
        or no meta.json at all) is accepted on the evidence of its arrays."""
        lens = {}
        for k, path in self._paths().items():
            if os.path.exists(path):
                a = np.load(path, mmap_mode="r")
                if a.dtype != self._dtype(k):
                    raise ValueError(f"{path} holds {a.dtype}, this sink writes {self._dtype(k)}")
                lens[k] = int(a.shape[0])
                del a
        if not lens:
            if meta_total:
                raise ValueError(f"{self.out_dir}: meta.json says {meta_total} rows but the arrays are missing")
            return 0
        if len(lens) != len(self.ARRAYS) or len(set(lens.values())) != 1:
            raise ValueError(f"{self.out_dir}: the merged arrays disagree ({lens}); refusing to merge on top of them")
        n = next(iter(lens.values()))
        if meta_total is not None and int(meta_total) != n:
            raise ValueError(f"{self.out_dir}: meta.json says {meta_total} rows, the arrays hold {n}")
        return n

    def _dtype(self, key):
        return self.pi_dtype if key == "mcts" else np.dtype(self.ARRAYS[key][1])

    def _write_dense_shard(self, base: str, s, p, w):
        np.save(base + "_s.npy", s.astype(np.float16, copy=False).reshape(-1, 17, 7, 10, 9))
        np.save(base + "_p.npy", p.astype(self.pi_dtype, copy=False).reshape(-1, 2086))
        np.save(base + "_z.npy", w.astype(np.float32, copy=False).reshape(-1))   # (written last: a shard counts once its z file exists)
        self._shards.append((base, int(len(w))))

    def _count_games(self, games: int):
        self.games += games
        if games:
            tmp = os.path.join(self.out_dir, "collect_state.json.tmp")
            with open(tmp, "w", encoding="utf-8") as f:
                json.dump({"iters": self.games}, f)
            os.replace(tmp, os.path.join(self.out_dir, "collect_state.json"))

    def append(self, states, pi, z, games: int = 1):
        to_np = lambda t: t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
        s, p, w = to_np(states), to_np(pi), to_np(z)
        if len(w):
            base = os.path.join(self.out_dir, f".shard_{os.getpid()}_{self._next:06d}")
            while os.path.exists(base + "_z.npy"):
                self._next += 1
                base = os.path.join(self.out_dir, f".shard_{os.getpid()}_{self._next:06d}")
            self._next += 1
            self._write_dense_shard(base, s, p, w)
        self._count_games(games)

    # ---- compact record shards (what the batched collector writes while it runs) ----------------------
    @staticmethod
    def _parse_rshard(path: str):
        """(path, plies, flags, plane_of_type) of a record shard, from its NAME (``.rshard_<pid>_<seq>_f<flags>_p<8 digits>.npy``)
        and its length; None for a file that is not one (e.g. a partial write: np.save goes through a temporary name)."""
        name = os.path.basename(path)
        try:
            parts = name[:-len(".npy")].split("_")
            flags = int(parts[-2][1:])
            pot = tuple(int(c) for c in parts[-1][1:])
            if not (parts[-2].startswith("f") and parts[-1].startswith("p") and len(pot) == 8):
                return None
            a = np.load(path, mmap_mode="r")
            if a.dtype != np.uint8 or a.ndim != 2 or a.shape[1] != 880:
                return None
            return (path, int(a.shape[0]), flags, pot)
        except Exception:
            return None

    def append_records(self, records, flags: int = 0, plane_of_type=None, games: int = 0):
        """Finished games as compact ply records: uint8 [P, 880] (torch or numpy; whole games, plies in order -- what
        ``engine.harvest_record_chunks`` / the exchange hand over). ``flags`` / ``plane_of_type`` = the engine's ``record_flags()`` /
        ``plane_of_type``: what :func:`engine.expand_records` needs to rebuild the rows at :meth:`finalize`."""
        rec = records.detach().cpu().numpy() if isinstance(records, torch.Tensor) else np.asarray(records)
        rec = np.ascontiguousarray(rec, dtype=np.uint8).reshape(-1, 880)
        pot = (0, 0, 1, 2, 3, 4, 5, 6) if plane_of_type is None else tuple(int(x) for x in plane_of_type)
        if len(pot) != 8 or max(pot) > 9 or min(pot) < 0:
            raise ValueError("plane_of_type must be 8 entries in 0..6")
        if len(rec):
            tail = f"_f{int(flags)}_p{''.join(str(x) for x in pot)}"
            path = os.path.join(self.out_dir, f".rshard_{os.getpid()}_{self._next:06d}{tail}.npy")
            while os.path.exists(path):
                self._next += 1
                path = os.path.join(self.out_dir, f".rshard_{os.getpid()}_{self._next:06d}{tail}.npy")
            self._next += 1
            tmp = path + ".tmp.npy"
            np.save(tmp, rec)
            os.replace(tmp, path)
            self._rshards.append((path, int(len(rec)), int(flags), pot))
        self._count_games(games)

    def _expand_record_shards(self):
        """Record shards -> dense shards, on the GPU (``ccz_expand_records``). A record shard is removed only after ALL its dense
        shards are written; dense shards carry its name (``.shard_r<tag>_k``), so an expansion that was interrupted is redone from
        the record shard without counting any row twice."""
        if not self._rshards:
            return
        if not torch.cuda.is_available():
            from ._lib import CczError
            raise CczError(f"{self.out_dir}: {len(self._rshards)} record shard(s) ({sum(r[1] for r in self._rshards)} plies) wait for the GPU expander "
                           "(ccz_expand_records): run finalize() where the collector ran; nothing was changed")
        from .engine import expand_records, game_aligned_chunks
        for path, plies, flags, pot in list(self._rshards):
            tag = os.path.basename(path)[len(".rshard_"):-len(".npy")]
            prefix = f".shard_r{tag}_"
            for name in os.listdir(self.out_dir):            # leftovers of an interrupted expansion of THIS shard
                if name.startswith(prefix):
                    os.remove(os.path.join(self.out_dir, name))
            self._shards = [(b, n) for b, n in self._shards if not os.path.basename(b).startswith(prefix)]
            rec = torch.from_numpy(np.load(path)).cuda()
            for k, part in enumerate(game_aligned_chunks(rec, 1 << 14)):   # bounds the dense temporary (2^15 rows = 1 GB)
                s, p, z = expand_records(part.contiguous(), flags, pot)
                self._write_dense_shard(os.path.join(self.out_dir, f"{prefix}{k:04d}"), s.cpu().numpy(), p.cpu().numpy(), z.cpu().numpy())
            os.remove(path)
            self._rshards.remove((path, plies, flags, pot))

    def rows(self) -> int:
        """Rows waiting in shards (not yet merged): dense shards + what the record shards expand to (two rows per ply with mirror images)."""
        from ._lib import FLAG_NO_MIRROR
        return int(sum(n for _, n in self._shards) + sum(p * (1 if f & FLAG_NO_MIRROR else 2) for _, p, f, _ in self._rshards))

    def finalize(self) -> int:
        """Merge what is on disk with the pending shards (the reference's converter step); returns the total row count.

        Crash-safe and idempotent: (1) every array is validated (dtype, length) before anything is written; (2) a journal
        naming ``n_old``, ``total`` and the shard files of THIS merge is written first and removed last, so a merge killed at
        any point -- also between ``meta.json`` and the deletion of its shards -- is completed by the next sink exactly once
        (:meth:`_recover`), never repeated; (3) ``meta.json`` changes after the arrays, the shards go after ``meta.json``."""
        self._expand_record_shards()
        n_old = self._rows_on_disk()
        if n_old != self._merged_rows:
            raise ValueError(f"{self.out_dir}: the arrays hold {n_old} rows, this sink merged {self._merged_rows}: another writer?")
        total = n_old + self.rows()
        if total == 0:
            return 0
        if not self._shards:
            self._write_meta(total)
            return total
        journal = {"n_old": int(n_old), "total": int(total), "pi_dtype": str(self.pi_dtype),
                   "shards": [[os.path.basename(b), int(n)] for b, n in self._shards]}
        tmp = self._journal() + ".tmp"
        with open(tmp, "w", encoding="utf-8") as f:
            json.dump(journal, f)
        os.replace(tmp, self._journal())
        self._apply(journal)
        self._shards = []
        self._merged_rows = total
        return total

    def _write_meta(self, total: int):
        meta = {"total_count": int(total),
                "states_shape": [int(total), 17, 7, 10, 9], "states_dtype": "float16",
                "mcts_shape": [int(total), 2086], "mcts_dtype": str(self.pi_dtype),
                "winners_shape": [int(total)], "winners_dtype": "float32",
                "iters": int(self.games)}
        tmp = os.path.join(self.out_dir, "meta.json.tmp")
        with open(tmp, "w", encoding="utf-8") as f:
            json.dump(meta, f, ensure_ascii=False, indent=2)  # convert.py:98-99
        os.replace(tmp, os.path.join(self.out_dir, "meta.json"))  # readers trust meta.json: it changes last

    def _apply(self, journal: dict):
        """Carry out (or complete) the merge a journal describes. Every step is skipped if already done."""
        n_old, total = int(journal["n_old"]), int(journal["total"])
        shards = [(os.path.join(self.out_dir, b), int(n)) for b, n in journal["shards"]]
        paths = self._paths()
        plan = {}
        for k, (suffix, _, tail) in self.ARRAYS.items():   # validate EVERYTHING before the first byte is written
            dtype = self._dtype(k)
            cur_n = 0
            if os.path.exists(paths[k]):
                cur = np.load(paths[k], mmap_mode="r")
                if cur.dtype != dtype:
                    raise ValueError(f"{paths[k]} holds {cur.dtype}, this sink writes {dtype}")
                cur_n = int(cur.shape[0])
                del cur
            if cur_n == total:
                plan[k] = False      # merged before the interruption
                continue
            if cur_n != n_old:
                raise ValueError(f"{paths[k]} has {cur_n} rows; the merge in progress expects {n_old} (before) or {total} (after)")
            for base, n in shards:
                if not os.path.exists(base + suffix):
                    raise ValueError(f"{base + suffix} is missing but {paths[k]} still needs it")
                a = np.load(base + suffix, mmap_mode="r")
                if int(a.shape[0]) != n or a.dtype != dtype or tuple(a.shape[1:]) != tuple(tail):
                    raise ValueError(f"{base + suffix}: {a.dtype}{a.shape}, expected {dtype}{(n,) + tuple(tail)}")
                del a
            plan[k] = True
        for k, (suffix, _, tail) in self.ARRAYS.items():
            if not plan[k]:
                continue
            tmp = paths[k] + ".tmp"
            out = np.lib.format.open_memmap(tmp, mode="w+", dtype=self._dtype(k), shape=(total,) + tail)
            if n_old:
                cur = np.load(paths[k], mmap_mode="r")
                out[:n_old] = cur[:n_old]
                del cur
            pos = n_old
            for base, n in shards:
                out[pos:pos + n] = np.load(base + suffix, mmap_mode="r")
                pos += n
            out.flush()
            del out
            os.replace(tmp, paths[k])
        self._write_meta(total)
        for base, _ in shards:
            for suffix, _, _ in self.ARRAYS.values():
                if os.path.exists(base + suffix):
                    os.remove(base + suffix)
        os.remove(self._journal())

    def _recover(self):
        """A journal on disk = a merge that did not finish: complete it (its shards are then gone before they could be
        adopted a second time)."""
        if not os.path.exists(self._journal()):
            return
        with open(self._journal(), encoding="utf-8") as f:
            journal = json.load(f)
        if journal.get("pi_dtype", str(self.pi_dtype)) != str(self.pi_dtype):
            raise ValueError(f"{self.out_dir}: an interrupted merge wrote mcts.npy as {journal['pi_dtype']}, this sink uses {self.pi_dtype}")
        log(f"completing the interrupted merge of {len(journal['shards'])} shards in {self.out_dir}", "WARNING")
        self._apply(journal)

    flush = finalize  # round-1 name


def write_games_hdf5(records: torch.Tensor, h5_path: str, flags: int = 0, plane_of_type=None, h5py_module=None) -> int:
    """The reference's OWN on-disk layout (collect.py:146-167): one group ``game_{i}`` per game in ``data.h5`` with datasets
    ``states`` (fp16 [2T,17,7,10,9], gzip), ``mcts_probs`` (float64 [2T,2086], gzip), ``winners`` (float64 [2T]) -- the T samples
    of the game followed by their T mirror images (``play_data + data_flip``, collect.py:131) -- and the file attribute ``iters``
    counting the games. Input: compact ply records (``harvest_record_chunks`` / ``RecordGatherer``: they carry the game
    boundaries the dense rows have lost); the rows are rebuilt on the GPU by ``ccz_expand_records``.

    ``h5py`` is not part of this image, so the collector's default sink stays ``TupleSink`` (the converter's .npy
    output, which is what the trainer reads); with h5py installed this function makes the GPU collector a drop-in producer for
    the reference's ``convert.py`` as well. ``h5py_module``: the module to use (tests pass a recording stand-in). Returns the
    new value of ``iters``."""
    if h5py_module is None:
        try:
            import h5py as h5py_module
        except ImportError as e:
            raise ImportError("write_games_hdf5 needs h5py (not installed in this image); TupleSink writes the converter's .npy "
                              "files instead") from e
    from .engine import expand_records, game_aligned_chunks, rows_of_records
    mul = rows_of_records(1, flags)
    with h5py_module.File(h5_path, "a") as h5f:
        it = int(h5f.attrs.get("iters", 0))
        for part in game_aligned_chunks(records, 1 << 13):
            part = part.contiguous()
            states, pi, z = (t.cpu().numpy() for t in expand_records(part, flags, plane_of_type))
            hdr = part[:, 96:100].cpu().numpy().view(np.uint16).reshape(-1, 2)   # (t, T) of every record
            p = 0
            while p < hdr.shape[0]:
                T = int(hdr[p, 1])
                lo, hi = mul * p, mul * (p + T)
                g = h5f.create_group(f"game_{it}")
                g.create_dataset("states", data=states[lo:hi], compression="gzip")
                g.create_dataset("mcts_probs", data=pi[lo:hi].astype(np.float64), compression="gzip")
                g.create_dataset("winners", data=z[lo:hi].astype(np.float64))
                it += 1
                h5f.attrs["iters"] = it
                p += T
    return it


class CollectPipeline:
    def __init__(self, init_model=None, n_boards: int = 1, n_playout: int = PLAYOUT, device: int = 0, seed: int = 0,
                 data_dir: str = DATA_DIR, reference_quirks: bool = False, num_channels: int = 256, resblocks_num: int = 40,
                 finalize_every: int = 0, on_playout=None, max_plies: int = 0, eval_cache_log2: int | None = None, gatherer=None,
                 dense_shards: bool = False):
        self.board = Board()                       # collect.py:28 (never advanced: source of the turn-plane quirk)
        self.game = Game(self.board, reference_quirks=reference_quirks)
        self.temp = 1.0
        self.n_playout = n_playout
        self.c_puct = C_PUCT
        self.init_model = init_model
        self.n_boards = n_boards
        self.device = device
        self.seed = seed
        self.reference_quirks = reference_quirks
        self._net_shape = (num_channels, resblocks_num)
        self.mcts_ai = None
        self.policy_value_net = None
        self.selfplay = None
        self.sink = TupleSink(data_dir)
        self.iters = self.sink.games
        self.episode_len = 0
        # batched path: evaluation cache of 2^n positions (528 B each: 2^24 = 8.9 GB of the GPU's 288); None = 24 from 192 boards on, else none
        self.eval_cache_log2 = eval_cache_log2
        self.max_plies = max_plies  # batched path: games adjudicated as draws at this many plies (0 = the engine's 2048)
        self.finalize_every = finalize_every
        self._finalized_at = self.sink.games
        self.on_playout = on_playout  # progress sink of the batched path (reference game.py:162-185 feeds a progress bar)
        # several ranks: the exchange of finished games (replay.AsyncRecordExchange / RecordGatherer / TupleGatherer); rank 0 stores the union
        self.gatherer = gatherer
        # batched path: False (default) = finished games go to disk as compact ply records and are expanded when the sink is finalized;
        # True = the dense rows themselves while collecting (rounds 1-4: 85 x the bytes through the host per move)
        self.dense_shards = bool(dense_shards)

    def load_model(self):
        """collect.py:48-62: load once; on failure fall back to a random-init net."""
        if self.policy_value_net is None:
            model_path = self.init_model if self.init_model else MODEL_DIR
            dev = f"cuda:{self.device}"
            try:
                self.policy_value_net = PolicyValueNet(model=model_path, device=dev, num_channels=self._net_shape[0],
                                                       resblocks_num=self._net_shape[1])
                log(f"Loaded model: {model_path}")
            except Exception as e:
                log(f"Failed to load model {model_path}: {e}", "ERROR")
                self.policy_value_net = PolicyValueNet(device=dev, num_channels=self._net_shape[0], resblocks_num=self._net_shape[1])
            if self.gatherer is not None and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
                # one job, one net: every rank plays with rank 0's weights (a model file that is missing on some rank, or the random
                # fallback above, must not give the ranks different evaluators) -- ONE broadcast of the fp32 state, before any exchange
                from .replay import broadcast_model
                broadcast_model(self.policy_value_net, src=0, what="state")
            self.mcts_ai = MCTS_AI(self.policy_value_net.policy_value_fn, c_puct=self.c_puct, n_playout=self.n_playout,
                                   is_selfplay=True, device=self.device, seed=self.seed)

    # ---- host path (one game at a time, the reference's own steps) ---------------------------------
    def preprocess(self, play_data):
        """collect.py:64-110"""
        processed = []
        self.episode_len = 0
        for i, (red_states, black_states, mcts_prob, winner) in enumerate(play_data):
            self.episode_len += 1
            if self.reference_quirks:
                red_turn = self.board.turn == RED          # collect.py:78: always RED
            else:
                red_turn = (i % 2 == 0)                   # side to move at ply i of a game that RED starts
            current_player = (np.ones if red_turn else np.zeros)((1, 7, 10, 9), dtype=np.float16)
            states = np.concatenate((red_states, black_states), axis=0)
            states = np.concatenate((states, current_player), axis=0)
            mcts_prob = np.asarray(mcts_prob)
            prob_sum = np.sum(mcts_prob)
            if prob_sum <= 0:
                log(f"mcts_prob sum is {prob_sum}; skipping this step", "WARNING")
                continue
            elif abs(prob_sum - 1.0) > 1e-6:
                mcts_prob = mcts_prob / prob_sum
            processed.append((states, mcts_prob, winner))
        return processed

    def flip_data(self, data):
        """collect.py:112-131: append the left-right mirror of every sample."""
        fm = flip_map()
        data_flip = []
        for states, mcts_prob, winner in data:
            states_flip = [np.flip(state, axis=2) for state in states]
            data_flip.append((states_flip, mcts_prob[fm], winner))
        return data + data_flip

    def collect_data(self, is_shown=False):
        """collect.py:133-176 (n_boards == 1) or one lockstep move of all boards + harvest (n_boards > 1)."""
        self.load_model()
        if self.n_boards > 1:
            return self.collect_batched(1, gatherer=self.gatherer)
        self.current_game_index = self.iters + 1
        play_data = self.game.start_self_play(self.mcts_ai, is_shown=is_shown, game_index=self.current_game_index)
        play_data = self.flip_data(self.preprocess(play_data))
        self.sink.append(np.array([np.asarray(s, dtype=np.float16) for s, _, _ in play_data]),
                         np.array([p for _, p, _ in play_data]), np.array([w for _, _, w in play_data]))
        self.iters = self.sink.games
        self._maybe_finalize()
        return self.iters

    def _maybe_finalize(self):
        """Rows stay in shards while collecting (the reference appends one group per game, collect.py:146-167, and converts
        offline); ``finalize_every`` > 0 merges them into the trainer's .npy files every that many games."""
        if self.finalize_every > 0 and self.sink.games - self._finalized_at >= self.finalize_every:
            if getattr(self, "gatherer", None) is not None and getattr(self.gatherer, "world", 1) > 1:
                # a merge on the launch thread of a rank whose peers expect its announcements (and, with a blocking gatherer, its
                # collective) within launch.DIST_TIMEOUT_S would make THEM fail: several ranks merge once, after the job's last exchange
                if not getattr(self, "_finalize_every_warned", False):
                    self._finalize_every_warned = True
                    log("finalize_every is ignored while an exchange with other ranks is live: the shards are merged once, at the end", "WARNING")
                return
            self.sink.finalize()
            self._finalized_at = self.sink.games

    # ---- MI355X path --------------------------------------------------------------------------------
    def collect_batched(self, n_moves: int, gatherer=None):
        """Play ``n_moves`` lockstep moves on all boards; harvest + (optionally) all-gather finished games."""
        from .selfplay import BatchedSelfPlay
        self.load_model()
        if self.selfplay is None:
            rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
            # compact evaluator boundary (logits in, the engine gathers the legal priors) and, on the fused evaluator path (>= 192
            # boards), the evaluation cache: positions evaluated before skip the network (same results)
            self.selfplay = BatchedSelfPlay(self.policy_value_net.evaluate_leaves_logits, self.n_boards, n_playout=self.n_playout,
                                            eval_cache_log2=(24 if self.n_boards >= 192 else 0) if self.eval_cache_log2 is None else int(self.eval_cache_log2),
                                            c_puct=self.c_puct, temp=self.temp, seed=self.seed, board_id_base=rank * self.n_boards,
                                            device=self.device, reference_quirks=self.reference_quirks, max_plies=self.max_plies)
            if getattr(self, "_viewer", None) is not None:
                self.selfplay.watch(0, self._viewer)
        for _ in range(n_moves):
            if hasattr(gatherer, "tick_until"):
                self._run_move_ticking(gatherer)
            else:
                self.selfplay.run_move(on_playout=self.on_playout)
            st = self.selfplay.engine.game_status()
            done = int(st["over"].sum())
            if gatherer is None:
                if done:
                    first = True
                    if self.dense_shards:   # round 1-4: the dense rows themselves (1.07 GB per move of 4096 boards through the host)
                        for chunk in self.selfplay.harvest_chunks(1 << 19):
                            self.sink.append(*chunk, games=done if first else 0)
                            first = False
                    else:                   # compact records (12.5 MB per move); TupleSink.finalize expands them on the GPU
                        e = self.selfplay.engine
                        for chunk in self.selfplay.harvest_record_chunks(1 << 16):
                            self.sink.append_records(chunk, e.record_flags(), e.plane_of_type, games=done if first else 0)
                            first = False
            elif hasattr(gatherer, "post"):
                # asynchronous exchange (replay.AsyncRecordExchange): this rank never waits for its peers -- the finished games join
                # its backlog, whatever exchange has completed meanwhile is stored; :meth:`drain_exchange` delivers the rest at the end
                chunks = list(self.selfplay.harvest_record_chunks(gatherer.cap)) if done else []
                for x in gatherer.post(chunks, games=done):
                    self._store_union(x.union, x.games, gatherer)
            elif hasattr(gatherer, "_payload"):
                # compact exchange (replay.RecordGatherer): finished games travel as 880-byte ply records, ONE collective per
                # move at the benchmark workload; rank 0 rebuilds the dense rows (ccz_expand_records) and stores the union
                from .replay import exchange_finished_games
                for union, games in exchange_finished_games(self.selfplay, gatherer, done):
                    self._store_union(union, games, gatherer)
            else:
                # Dense exchange (replay.TupleGatherer, round 2's wire format). Every rank calls gather() the same number of
                # times: once per move at least (possibly with zero rows), and again while ANY rank still holds harvest chunks.
                e = self.selfplay.engine
                it = iter(self.selfplay.harvest_chunks(gatherer.cap)) if done else iter(())
                chunk = next(it, None)
                first = True
                while True:
                    nxt = next(it, None) if chunk is not None else None
                    if chunk is None:
                        chunk = (e.leaf_input[:0], torch.empty((0, 2086), device=e.device), torch.empty((0,), device=e.device))
                    s, p, z = gatherer.gather(*(t.to(gatherer.device) for t in chunk), more=nxt is not None, user=done if first else 0)
                    if gatherer.rank == 0:  # the union of the shards goes to ONE store, as N collectors -> one data file
                        self.sink.append(s, p, z, games=gatherer.user_sum)
                    else:
                        self.sink.games += gatherer.user_sum
                    first = False
                    if not gatherer.any_more:
                        break
                    chunk = nxt
            self.iters = self.sink.games
            self._maybe_finalize()
        self.selfplay.engine.check_healthy()
        return self.iters

    def _run_move_ticking(self, gatherer):
        """One move of all boards with the asynchronous exchange kept moving meanwhile, as bench.py does it: ``tick()`` from the search's
        throttled ``on_playout`` callback (every n/100 simulations: an exchange every rank has announced is issued within a few steps,
        not at this rank's next move boundary -- its kernel would spin on the peers' CUs until then) and ``tick_until`` while the host
        waits for the GPU in front of the move. An exception raised by the exchange inside the callback (a peer's abort, the
        announcement timeout) is re-raised here: the search swallows what its callback raises (mcts.py:156-159)."""
        sp = self.selfplay
        failed = []

        def store(done):
            for x in done:
                self._store_union(x.union, x.games, gatherer)

        def on_playout(k):
            if self.on_playout is not None:
                try:
                    self.on_playout(k)
                except Exception:
                    pass
            if not failed:
                try:
                    store(gatherer.tick())
                except BaseException as exc:   # noqa: BLE001  (kept for the caller: see above)
                    failed.append(exc)

        def boundary():
            dev = sp.engine.device
            if not failed and dev.type == "cuda":
                caught_up = torch.cuda.Event()
                caught_up.record(torch.cuda.current_stream(dev))
                while not caught_up.query():
                    store(gatherer.tick_until(caught_up))
            return sp.finish_move()

        moves = sp.advance(sp.n_playout - sp._sim, on_playout=on_playout, boundary=boundary)
        if failed:
            raise failed[0]
        return moves

    def _store_union(self, union, games: int, gatherer):
        """Records of ALL ranks' finished games (one exchange): rank 0 rebuilds the dense rows (ccz_expand_records) and appends them
        to ONE store, as N reference collectors appending to one data file would (collect.py:146-167); the other ranks count the games."""
        from .engine import expand_records, game_aligned_chunks
        e = self.selfplay.engine
        if gatherer.rank == 0:
            if not self.dense_shards:   # the records as they are (880 B per ply); TupleSink.finalize expands them on the GPU
                self.sink.append_records(union, e.record_flags(), e.plane_of_type, games=games)
            else:
                first = True
                for part in game_aligned_chunks(union.to(e.device), 1 << 14):  # bounds the dense temporary (2^15 rows = 1 GB)
                    self.sink.append(*expand_records(part.contiguous(), e.record_flags(), e.plane_of_type), games=games if first else 0)
                    first = False
                if first:
                    self.sink.append(e.leaf_input[:0], torch.empty((0, 2086)), torch.empty((0,)), games=games)
        else:
            self.sink.games += games
        self.iters = self.sink.games

    def drain_exchange(self, gatherer):
        """End of a multi-rank collection with an asynchronous exchange: blocking, every rank calls it; afterwards every record of
        every rank has reached rank 0's store."""
        for x in gatherer.flush_iter():
            self._store_union(x.union, x.games, gatherer)
        self._maybe_finalize()
        return self.iters

    def run(self, is_shown=False, max_calls: int = 0, viewer=None, finalize: bool = True):
        """collect.py:178-186: collect until interrupted (``max_calls`` > 0 stops after that many ``collect_data`` calls:
        games on the single-board path, lockstep moves on the batched one). ``is_shown`` (reference ``--show``) pushes the single
        game, or board 0 of the batch, to ``viewer`` -- anything with ``update_board(svg, status)``; the HTTP window itself is not
        part of the package (``examples/viewer.py``).

        The end of a run. Normal end or Ctrl-C: the exchange is drained (blocking, every rank), then the shards are merged into the
        trainer's files -- unless ``finalize=False``: a multi-rank job merges AFTER its process group is gone (the CLI below), because
        rank 0's merge takes minutes for a long collection while its peers would sit in a barrier under the group's 180-s timeout.
        An exception (a peer's abort, the announcement timeout, an engine error flag): no drain -- it would block for another timeout
        on peers that will never announce and bury the first error under a second one -- and, with other ranks in the job, no merge:
        the error goes straight up (``launch.guarded`` ends the process, the launcher the job); the shards stay on disk and the next
        sink adopts them."""
        if is_shown and self.n_boards > 1:
            if viewer is None:
                log("--show without a viewer: pass viewer=... (e.g. examples/viewer.py get_chess_window()); nothing is displayed", "WARNING")
            self._viewer = viewer
        elif is_shown and viewer is not None and getattr(self, "game", None) is not None:
            self.game.viewer = viewer   # the one-game-at-a-time loop: Game.graphic pushes every position (game.py:47-75)
        calls = 0
        multi = self.gatherer is not None and getattr(self.gatherer, "world", 1) > 1
        try:
            try:
                while max_calls <= 0 or calls < max_calls:
                    iters = self.collect_data(is_shown=is_shown)
                    calls += 1
                    log(f"Episode {iters}, steps {self.episode_len}")
            except KeyboardInterrupt:
                log("Exit")
            if self.gatherer is not None and hasattr(self.gatherer, "flush_iter") and self.selfplay is not None:
                self.drain_exchange(self.gatherer)   # every rank's last games reach rank 0's store (blocking; all ranks call it)
        except BaseException:
            if not multi:   # one process, nobody waits for it: what was collected is merged before the error goes up
                try:
                    self.sink.finalize()
                except Exception as exc:   # (the first error is the one to report; the shards stay on disk for the next sink)
                    log(f"merging the shards after an error failed as well: {exc}", "WARNING")
            raise
        if finalize:
            self.sink.finalize()


if __name__ == "__main__":
    import argparse
    parser = argparse.ArgumentParser(description="collect Xiangqi self-play data on MI355X")
    parser.add_argument("--show", action="store_true", default=False)
    parser.add_argument("--model", type=str, default="current_policy.pkl")
    parser.add_argument("--boards", type=int, default=4096, help="concurrent boards on this GPU (1 = the reference's one-game-at-a-time loop)")
    parser.add_argument("--playout", type=int, default=PLAYOUT)
    parser.add_argument("--moves", type=int, default=0, help="stop after this many collect_data calls (0 = until interrupted, as the reference)")
    parser.add_argument("--max-plies", type=int, default=0, help="batched path: adjudicate games at this many plies (0 = 2048)")
    parser.add_argument("--eval-cache-log2", type=int, default=None, help="batched path: evaluation cache of 2^n positions, 528 B each (0 = none; default 24 = 8.9 GB from 192 boards on)")
    parser.add_argument("--data-dir", type=str, default=DATA_DIR)
    parser.add_argument("--channels", type=int, default=256)
    parser.add_argument("--blocks", type=int, default=40)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--backend", default="nccl", help="several ranks (under torch.distributed.run): nccl = RCCL, one GPU per rank; gloo to rehearse")
    parser.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0 (with --backend gloo)")
    args = parser.parse_args()
    # N collectors as ONE job (the reference starts N shell commands, README.md:31-48): python -m torch.distributed.run --nproc-per-node N
    # -m chinesechesszero_amd.collect ... -- every rank plays --boards boards, rank 0 stores the union of the finished games
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    gatherer, device = None, 0
    if world > 1:
        from . import launch
        from .replay import AsyncRecordExchange
        device = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(device)
        launch.init_distributed(args.backend, torch.device("cuda", device))
        gatherer = AsyncRecordExchange(max(32768, args.max_plies or 2048), torch.device("cuda", device) if args.backend == "nccl" else "cpu")
        if rank > 0:   # these ranks store nothing (their sink only counts games): keep them out of rank 0's directory lock
            args.data_dir = os.path.join(args.data_dir, f".rank{rank}")
    viewer = None
    if args.show:   # the window is an example, not part of the package: found when run from a checkout of the repository
        try:
            import sys
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from examples.viewer import get_chess_window
            viewer = get_chess_window()
        except Exception as exc:
            log(f"--show: examples/viewer.py is not importable here ({exc}); running without a window", "WARNING")
    pipe = CollectPipeline(init_model=args.model, n_boards=args.boards, n_playout=args.playout, data_dir=args.data_dir, seed=args.seed,
                           num_channels=args.channels, resblocks_num=args.blocks, max_plies=args.max_plies, device=device,
                           eval_cache_log2=args.eval_cache_log2, gatherer=gatherer)
    if world > 1:
        from .launch import guarded

        def _job():
            # the merge (rank 0 expands every record shard on the GPU -- 85x the bytes -- and writes the trainer's files) runs AFTER the
            # group is gone: under the barrier it kept the peers waiting past the group's timeout for any collection of more than
            # ~15 moves, and the launcher then killed rank 0 in the middle of it
            pipe.run(is_shown=args.show and rank == 0, max_calls=args.moves, viewer=viewer, finalize=False)
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
            pipe.sink.finalize()
            return 0
        raise SystemExit(guarded(_job))
    pipe.run(is_shown=args.show, max_calls=args.moves, viewer=viewer)
