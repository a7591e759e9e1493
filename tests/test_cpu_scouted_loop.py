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


class _FakeEngine:
    """What ScoutedSearch.run asks of an engine, scripted: each scouted run returns the next (simulations done, evaluator needed) pair."""
    B, n_scouts, leaf_input = 11, 10, "leaf rows"

    def __init__(self, first_state, outcomes):
        import torch
        self.device = torch.device("cpu")
        self.first_state, self.outcomes, self.log = first_state, list(outcomes), []

    def select_leaves(self):
        self.log.append("select")

    def scout_and_plan(self):
        self.log.append("plan")

    def plan_state_of_board0(self):
        return self.first_state

    def set_run(self, budget, left):
        assert 1 <= budget <= left
        self.log.append(("set_run", budget, left))

    def scouted_run_launch(self):
        self.log.append("run")

    def run_outcome(self):
        return self.outcomes.pop(0)

    def gather_priors_planned(self, logits, value):
        self.log.append(("gather", logits, value))

    def clear_eval_cache(self):
        self.log.append("clear")


def test_the_evaluator_runs_exactly_when_the_last_run_ended_on_a_miss():
    """ScoutedSearch.run: evaluator + gather in front of the run iff the previous run (or begin_move's plan) said board 0's leaf is not
    in the table; the budget handed to the engine never exceeds what the move has left; a run that reports no simulation is an error."""
    from chinesechesszero_amd.selfplay import ScoutedSearch
    calls = []

    def evaluator(leaf):
        calls.append(leaf)
        return "logits", "value"
    evaluator.batched = evaluator.returns_logits = evaluator.stateless = True
    e = _FakeEngine(first_state=0, outcomes=[(1, True), (7, False), (2, True), (5, False)])
    s = ScoutedSearch(e, evaluator, use_graph=True, device_loop=True)     # (graphs need a GPU: use_graph falls back to eager on this engine)
    assert s.device_loop and not s.use_graph
    s.begin_move()
    assert e.log == ["select", "plan"] and s._need
    left = 15
    for budget in (15, 14, 4, 5):
        left -= s.run(left, budget)
    assert left == 0 and s.simulations == 15 and s.evaluator_calls == 3 and calls == ["leaf rows"] * 3
    assert [x for x in e.log if isinstance(x, tuple) and x[0] == "set_run"] == [("set_run", 15, 15), ("set_run", 14, 14), ("set_run", 4, 7), ("set_run", 5, 5)]
    # the evaluator (and the gather of its rows) precedes runs 1, 2 and 4: begin_move's plan and runs 1 and 3 ended on a miss, run 2 on its budget
    gathers_before_run = []
    pending = False
    for x in e.log:
        if isinstance(x, tuple) and x[0] == "gather":
            pending = True
        elif x == "run":
            gathers_before_run.append(pending)
            pending = False
    assert gathers_before_run == [True, True, False, True]
    # a table hit at the start of a move: the first run needs no evaluator
    e2 = _FakeEngine(first_state=1, outcomes=[(3, False), (0, False)])
    s2 = ScoutedSearch(e2, evaluator, use_graph=False, device_loop=True)
    s2.begin_move()
    assert not s2._need and s2.run(10, 10) == 3 and s2.evaluator_calls == 0
    with pytest.raises(RuntimeError):
        s2.run(7, 7)
