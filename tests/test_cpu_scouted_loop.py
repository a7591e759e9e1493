"""CPU: the host side of the one-game path's device loop (mcts.MCTS.get_move_probs over selfplay.ScoutedSearch.run).

The device does the simulations in runs of unknown length (a run ends at a table miss, at the caller's budget or at the move's end); the
host must still call ``on_playout`` at exactly the playouts -- and with exactly the counts -- of the reference's loop (mcts.py:150-160:
every ``max(1, n_playout // 100)`` playouts and after the last one), and must never hand the device a budget that would run past such
a playout. No GPU: the engine and the search are stand-ins that record what they are asked for."""
import numpy as np
import pytest


class _Runs:
    """Stands in for ScoutedSearch: ``run`` does min(budget, a random run length, left) simulations."""
    device_loop = True

    def __init__(self, seed):
        self.rng = np.random.RandomState(seed)
        self.asked = []
        self.began = 0

    def begin_move(self):
        self.began += 1

    def run(self, left, budget):
        assert 1 <= budget <= left, (left, budget)
        done = int(min(budget, left, 1 + self.rng.randint(0, 40)))
        self.asked.append((left, budget, done))
        return done


class _Engine:
    def root_children(self):
        return {"k": np.array([2]), "acts": np.array([[7, 9]], np.uint16), "visits": np.array([[3, 1]], np.int32)}

    def check_healthy(self):
        pass


def _reference_reports(n):
    """What mcts.py:150-160 reports for n playouts."""
    interval, acc, out = max(1, n // 100), 0, []
    for i in range(n):
        acc += 1
        if acc >= interval or i == n - 1:
            out.append(acc)
            acc = 0
    return out


@pytest.mark.parametrize("n", [1, 2, 37, 200, 250, 1600, 1999])
def test_on_playout_is_called_where_the_reference_calls_it(n, monkeypatch):
    from chinesechesszero_amd import mcts as M

    def evaluator(leaf):
        raise AssertionError("the stand-in search never evaluates")
    evaluator.batched = True
    evaluator.returns_logits = True
    m = M.MCTS(evaluator, 5, n, scouts=10)
    assert m.scouts == 10
    m._engine, m._scouted = _Engine(), _Runs(n)
    monkeypatch.setattr(m, "_sync_root", lambda board: None)
    seen = []
    acts, probs = m.get_move_probs(None, temp=1.0, on_playout=seen.append)
    assert seen == _reference_reports(n) and sum(seen) == n
    assert sum(d for _, _, d in m._scouted.asked) == n and m._scouted.began == 1
    interval = max(1, n // 100)
    assert all(b <= interval for _, b, _ in m._scouted.asked)      # a run never crosses a playout on_playout is due at
    assert acts == (7, 9) and abs(float(probs.sum()) - 1.0) < 1e-12
    # without a callback the device may run to the end of the move: the budget is everything that is left
    m._scouted = _Runs(n + 1)
    m.get_move_probs(None, temp=1.0)
    assert all(b == left for left, b, _ in m._scouted.asked) and sum(d for _, _, d in m._scouted.asked) == n
    # a callback that raises does not stop the search (mcts.py:156-159)
    m._scouted = _Runs(n + 2)

    def boom(k):
        raise RuntimeError("viewer went away")
    m.get_move_probs(None, temp=1.0, on_playout=boom)
    assert sum(d for _, _, d in m._scouted.asked) == n
