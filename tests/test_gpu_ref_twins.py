"""GPU: libcczero.so against the host twins of the same entry points (oracle/ccz_ref.c: ``ccz_ref_*`` = the signatures of
include/cczero.h on host memory). ONE driver function runs a script of lockstep simulations and moves on either
backend -- same calls, same order, same evaluator numbers -- and everything the two return is compared bit for bit."""
import numpy as np
import pytest
import torch

from gpu_harness import make_evaluator, planes_to_squares

pytestmark = pytest.mark.gpu


class Device:
    """SelfPlayEngine behind the call surface both backends share (NumPy in, NumPy out)."""

    def __init__(self, B, **kw):
        from chinesechesszero_amd.engine import SelfPlayEngine
        self.e = SelfPlayEngine(B, strict=True, **kw)
        self.B = B

    def set_position(self, *a):
        self.e.set_position(*a)

    def select_leaves(self):
        return self.e.select_leaves().cpu().numpy()

    def expand_backup(self, P, V):
        self.e.expand_backup(torch.from_numpy(P).to(self.e.device), torch.from_numpy(V).to(self.e.device))

    def finish_move(self, forced_moves=None, temps=None, keep_tree=True):
        return self.e.finish_move(forced_moves=forced_moves, temps=temps, keep_tree=keep_tree).cpu().numpy().copy()

    def root_children(self):
        return self.e.root_children()

    def game_status(self):
        return self.e.game_status()

    def leaf_info(self):
        return self.e.leaf_info()

    def root_positions(self):
        return self.e.root_positions()

    def close(self):
        self.e.check_healthy()
        self.e.close()


def drive(x, kind, salts, n, moves, starts=None, forced_at=()):
    """The script: per move n simulations (select -> evaluator -> expand+backup), the roots, then the move (sampled on the per-board
    Philox stream, or forced to the first child on the moves listed in ``forced_at``). Returns everything observed."""
    ev = make_evaluator(kind, salts)
    log = []
    for b, st in enumerate(starts or []):
        if st is not None:
            x.set_position(b, *st)
    for mv in range(moves):
        for s in range(n):
            planes = x.select_leaves().astype(np.float32)
            info = {k: v.copy() for k, v in x.leaf_info().items()}
            none = info["status"] == 3                       # finished boards select nothing: whatever their slots still hold is not compared
            planes[none] = 0.0
            info["k"][none] = 0
            info["depth"][none] = 0
            info["ids"][np.arange(128)[None, :] >= info["k"][:, None]] = 0
            log.append(("leaf", planes.copy(), info))
            sq, turn = planes_to_squares(planes)
            P, V = ev(sq, turn)
            x.expand_backup(P, V)
        rc = x.root_children()
        over = x.game_status()["over"].astype(bool)
        log.append(("roots", {k: np.where(over.reshape(-1, *([1] * (v.ndim - 1))), 0, v) for k, v in rc.items()}))
        forced = None
        if mv in forced_at:
            forced = np.where(over | (rc["k"] == 0), -1, rc["acts"][:, 0].astype(np.int32)).astype(np.int32)
        log.append(("moves", x.finish_move(forced_moves=forced)))
        st = {k: v.copy() for k, v in x.game_status().items()}
        st["winner"][st["over"] == 0] = -1
        log.append(("status", st, x.root_positions()))
    return log


def same(a, b, where=""):
    assert type(a) is type(b), where
    if isinstance(a, dict):
        assert sorted(a) == sorted(b), where
        for k in a:
            same(a[k], b[k], f"{where}.{k}")
    elif isinstance(a, (tuple, list)):
        assert len(a) == len(b), where
        for i, (u, v) in enumerate(zip(a, b)):
            same(u, v, f"{where}[{i}]")
    elif isinstance(a, np.ndarray):
        assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes(), where
    else:
        assert a == b, where


@pytest.mark.parametrize("kind,n,moves", [("hash", 40, 4), ("hash_sharp", 64, 3), ("uniform", 24, 2)])
def test_the_library_and_its_host_twins_return_the_same_bytes(kind, n, moves):
    from golden_cases import STARTS
    from oracle import RefEngine
    B = 6
    salts = [3, 4, 5, 6, 7, 8]
    starts = [None, None, (STARTS["two_rooks"].copy(), 1, 0), (STARTS["rook_knight"].copy(), 0, 3), (STARTS["pawns"].copy(), 1, 110), None]
    kw = dict(n_playout=n, seed=9, board_id_base=1000, eps=0.25, alpha=0.2, temp=1.0)
    dev, ref = Device(B, **kw), RefEngine(B, **kw)
    try:
        a = drive(dev, kind, salts, n, moves, starts, forced_at=(1,))
        b = drive(ref, kind, salts, n, moves, starts, forced_at=(1,))
        assert len(a) == len(b) == moves * (n + 3)
        for i, (u, v) in enumerate(zip(a, b)):
            same(u, v, f"event {i} ({u[0]})")
    finally:
        ref.close()
        dev.close()
