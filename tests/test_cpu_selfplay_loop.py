"""CPU: the ONE lockstep loop (BatchedSelfPlay.advance) that run_move, bench.py, the soak and the GPU tests all run, against a
recording stand-in for the engine: launch sequence of a move (reference mcts.py:150-160 loop body, batched), hook order, the
boundary callback, search() stopping in front of the move, the planned (evaluation cache) branch and its weights-version check."""
import numpy as np
import pytest
import torch


class FakeEngine:
    def __init__(self, n_boards, n_playout=400, eval_cache_log2=0, **kw):
        self.B, self.n_playout, self.eval_cache_log2 = n_boards, n_playout, eval_cache_log2
        self.device = torch.device("cpu")
        self.leaf_input = torch.zeros((n_boards, 17, 7, 10, 9), dtype=torch.float16)
        self.calls = []
        self.max_plies = 2048

    def _rec(self, name):
        self.calls.append(name)

    def select_leaves(self):
        self._rec("select")
        return self.leaf_input

    def step(self, prob, value):
        self._rec("step")
        return self.leaf_input

    def expand_backup(self, prob, value):
        self._rec("expand_backup")

    def gather_priors(self, logits, value):
        self._rec("gather")

    def step_compact(self, value):
        self._rec("step_compact" if value is not None else "step_compact(engine values)")
        return self.leaf_input

    def expand_backup_compact(self, value):
        self._rec("expand_backup_compact" if value is not None else "expand_backup_compact(engine values)")

    def eval_plan(self):
        self._rec("plan")
        return torch.zeros(self.B, dtype=torch.int32), torch.zeros(1, dtype=torch.int32)

    def gather_priors_planned(self, logits, value):
        self._rec("gather_planned")

    def clear_eval_cache(self):
        self._rec("clear_cache")

    def finish_move(self, *a, **k):
        self._rec("finish_move")
        return torch.zeros(self.B, dtype=torch.int32)


@pytest.fixture
def fake(monkeypatch):
    from chinesechesszero_amd import selfplay
    monkeypatch.setattr(selfplay, "SelfPlayEngine", FakeEngine)
    return selfplay


def dense(leaf):
    return torch.zeros((leaf.shape[0], 2086)), torch.zeros(leaf.shape[0])


def test_a_move_is_select_then_n_minus_1_fused_steps_then_expand_backup_then_the_move(fake):
    sp = fake.BatchedSelfPlay(dense, 3, n_playout=4)
    moves = sp.run_move()
    assert moves is not None
    assert sp.engine.calls == ["select", "step", "step", "step", "expand_backup", "finish_move"]
    # the next move starts with a fresh selection; advance() may stop and resume anywhere inside a move
    sp.engine.calls.clear()
    assert sp.advance(2) is None and sp._sim == 2
    assert sp.engine.calls == ["select", "step", "step"]
    sp.advance(3)                                   # finishes the move (2 more simulations) and starts the next one
    assert sp.engine.calls == ["select", "step", "step", "step", "expand_backup", "finish_move", "select", "step"] and sp._sim == 1


def test_hooks_boundary_and_search(fake):
    class Logits:
        returns_logits = True

        def __call__(self, leaf):
            return torch.zeros((leaf.shape[0], 2086)), torch.zeros(leaf.shape[0])
    sp = fake.BatchedSelfPlay(Logits(), 2, n_playout=3)
    seen, played = [], []
    sp.advance(3, hooks=lambda stage, i: seen.append((stage, i)), boundary=lambda: played.append("boundary") or sp.finish_move())
    assert seen == [("eval0", 0), ("eval1", 0), ("step1", 0), ("eval0", 1), ("eval1", 1), ("step1", 1), ("eval0", 2), ("eval1", 2), ("step1", 2)]
    assert played == ["boundary"]
    assert sp.engine.calls == ["select", "gather", "step_compact", "gather", "step_compact", "gather", "expand_backup_compact", "finish_move"]
    # search(): all simulations of the move, the move itself is NOT played (the roots can be read first)
    sp.engine.calls.clear()
    sp.search()
    assert sp.engine.calls[-1] == "expand_backup_compact" and "finish_move" not in sp.engine.calls and sp._sim == 0
    sp.finish_move()
    # the reference's throttled progress callback (mcts.py:153-160): swallowed exceptions, the count adds up
    got = []
    sp.advance(3, on_playout=lambda k: got.append(k) or (_ for _ in ()).throw(RuntimeError("ignored")))
    assert sum(got) == 3


def test_planned_branch_and_the_weights_version_check(fake):
    class Owner:
        weights_version = 0

        def ev(self, leaf, plan=None):
            return torch.zeros((leaf.shape[0], 2086), dtype=torch.float16), torch.zeros(leaf.shape[0])
    o = Owner()
    o.ev.__func__.returns_logits = True
    o.ev.__func__.accepts_plan = True
    sp = fake.BatchedSelfPlay(o.ev, 2, n_playout=2, eval_cache_log2=12)
    assert sp.planned
    sp.run_move()
    assert sp.engine.calls == ["select", "plan", "gather_planned", "step_compact(engine values)", "plan", "gather_planned",
                               "expand_backup_compact(engine values)", "finish_move"]
    sp.engine.calls.clear()
    o.weights_version = 1                           # new weights: the table is emptied before the next evaluation
    sp.advance(1)
    assert sp.engine.calls[:3] == ["select", "clear_cache", "plan"]
    # an evaluator that accepts a plan but has nothing that says when its weights change is refused
    class Bare:
        returns_logits = True
        accepts_plan = True

        def __call__(self, leaf, plan=None):
            return None
    with pytest.raises(ValueError):
        fake.BatchedSelfPlay(Bare(), 2, n_playout=2, eval_cache_log2=12)
    b = Bare()
    b.stateless = True
    assert fake.BatchedSelfPlay(b, 2, n_playout=2, eval_cache_log2=12).planned


def test_eval_options_are_read_once_and_changed_with_set_options(monkeypatch):
    from chinesechesszero_amd.net import EvalOptions, InferenceNet, Net
    monkeypatch.setenv("CCZ_CONV_LAYOUT", "g16")
    monkeypatch.setenv("CCZ_TOWER_CHAINS", "4")
    monkeypatch.setenv("CCZ_CONV_EDGE_TILES", "1")
    inf = InferenceNet(Net(256, 1))
    monkeypatch.delenv("CCZ_CONV_LAYOUT")           # too late to matter: read when the object was built
    assert inf.opt.layout == "g16" and inf.opt.chains == 4 and inf.opt.edge_tiles is True
    assert inf.opt.non_default() == {"layout": "g16", "chains": 4, "edge_tiles": True}
    assert inf._g16(100) and not inf._g16(64) and inf._edge(100)
    inf.set_options(layout="auto", chains=0, edge_tiles="auto")
    assert not inf._g16(100) and inf._g16(640) and inf.tower_chains(4096, 1, inf._edge(4096)) == 3 and inf.tower_chains(4096) == 2
    assert inf._edge(4096) and not inf._edge(3072) and not inf._edge(4096, g16=False)
    with pytest.raises(KeyError):
        inf.set_options(no_such_switch=1)
    with pytest.raises(ValueError):
        EvalOptions(env={"CCZ_CONV_LAYOUT": "nchw"})
    # one cached decision per batch shape: a CPU tensor never takes the hand-written kernels
    x = torch.zeros((2, 17, 7, 10, 9), dtype=torch.float16)
    assert inf._path(x) == "torch"
