"""GPU: the evaluation cache (ccz_eval_plan / ccz_gather_priors_planned, include/cczero.h). The reference evaluates every
leaf (mcts.py:114); here a position evaluated before -- by this board, another board, or another board of the same step -- is
served from a table keyed by the leaf's Zobrist key, and only the remaining rows go through the evaluator. The bar: NOTHING
changes -- visit counts, Q, priors, the moves played -- against the same engine without a cache."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class LogitsEvaluator:
    """A deterministic device-side 'net' that returns LOGITS (compact boundary) and accepts a plan: a pure function of the
    position, so a cached result is the recomputed result. Counts the rows it was asked for."""
    batched = True
    returns_logits = True
    accepts_plan = True
    stateless = True   # fixed weights: nothing can invalidate a cached evaluation

    def __init__(self, device, seed=0, sharp=6.0):
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.W = (torch.randn(1890, 2086, generator=g) * sharp / 5.6).to(device)
        self.w = (torch.randn(1890, generator=g) * 0.7).to(device)
        self.rows_asked = 0
        self.calls = 0

    def __call__(self, leaf, plan=None):
        B = leaf.shape[0]
        if plan is not None:
            rows, n = plan
            self.rows_asked += int(n.item())
            leaf = leaf.index_select(0, rows.long().clamp(0, B - 1))     # compact: row i = board rows[i]
        else:
            self.rows_asked += B
        self.calls += 1
        x = leaf.view(B, 17, 630)
        x = torch.cat([x[:, 7], x[:, 15], x[:, 16]], dim=1).float()
        return (x @ self.W).contiguous(), torch.tanh(x @ self.w).contiguous()


def _play(B, n, moves, cache_log2, dev, seed=3, max_plies=40, **engine_kw):
    from chinesechesszero_amd.selfplay import BatchedSelfPlay
    ev = LogitsEvaluator(dev, seed=1)
    sp = BatchedSelfPlay(ev, B, n_playout=n, seed=seed, max_plies=max_plies, eval_cache_log2=cache_log2, **engine_kw)
    assert sp.planned == (cache_log2 > 0)
    trace = []
    for _ in range(moves):
        # search only (the move is played below, after the roots have been read)
        e = sp.engine
        leaf = e.select_leaves()
        for i in range(n):
            if sp.planned:
                lg, v = sp._planned_eval(leaf)
                if i + 1 < n:
                    leaf = e.step_planned(lg, v)
                else:
                    e.expand_backup_planned(lg, v)
            else:
                lg, v = ev(leaf)
                if i + 1 < n:
                    leaf = e.step_logits(lg, v)
                else:
                    e.expand_backup_logits(lg, v)
        rc = e.root_children()
        mv = sp.finish_move().cpu().numpy().copy()
        trace.append((rc, mv))
        if e.game_status()["over"].any():
            sp.harvest()
    sp.engine.check_healthy()
    return sp, ev, trace


def test_cached_search_is_bit_identical_and_skips_repeated_positions():
    dev = torch.device("cuda", 0)
    B, n, moves = 48, 40, 6
    sp0, ev0, t0 = _play(B, n, moves, 0, dev)
    sp1, ev1, t1 = _play(B, n, moves, 14, dev)
    for (a, ma), (b, mb) in zip(t0, t1):
        for key in ("k", "acts", "visits", "root_visits"):
            assert np.array_equal(a[key], b[key]), key
        assert np.array_equal(a["q"].view(np.uint32), b["q"].view(np.uint32)) and np.array_equal(a["prior"].view(np.uint32), b["prior"].view(np.uint32))
        assert np.array_equal(ma, mb)
    s0, s1 = sp0.engine.stats(), sp1.engine.stats()
    for key in ("sims", "moves", "games", "expansions", "terminal_leaves", "sum_depth", "sum_children"):
        assert s0[key] == s1[key], key
    assert s0["cache_probes"] == 0 and s1["cache_probes"] == s1["expansions"]
    # all 48 boards start on the opening position and search it identically in their first move: one row serves 48 boards
    assert s1["cache_shared_rows"] >= 47 * (n - 1) and s1["cache_hits"] > 0 and s1["cache_stores"] > 0
    computed = s1["cache_probes"] - s1["cache_hits"] - s1["cache_shared_rows"]
    assert ev1.rows_asked == computed and ev0.rows_asked == B * n * moves
    assert computed < 0.8 * s1["cache_probes"]


@pytest.mark.parametrize("variant", ["python-chess-lineage", "value_f16"])
def test_cached_search_under_another_rule_preset_and_with_the_float16_value(variant):
    """The cache under the options that change what an entry means or how it is consumed: a rule preset with another `legal_moves`
    order (the priors of an entry are stored IN that order, and its 24-bit tag is a hash of that ordered list) and another plane
    numbering; the float16 value arithmetic of the reference's CUDA path (the table keeps the float32 value, the backup rounds).
    Cached and uncached searches stay identical."""
    from chinesechesszero_amd import tools
    dev = torch.device("cuda", 0)
    kw = {"value_f16": True} if variant == "value_f16" else {}
    try:
        if variant != "value_f16":
            tools.set_rules(preset=variant)
        sp0, _, t0 = _play(40, 36, 5, 0, dev, seed=11, **kw)
        sp1, _, t1 = _play(40, 36, 5, 14, dev, seed=11, **kw)
    finally:
        tools.set_rules()
    if variant != "value_f16":
        assert sp1.engine.move_rank is not None and sp1.engine.type_rank is not None
    for (a, ma), (b, mb) in zip(t0, t1):
        for key in ("k", "acts", "visits", "root_visits"):
            assert np.array_equal(a[key], b[key]), key
        assert np.array_equal(a["q"].view(np.uint32), b["q"].view(np.uint32)) and np.array_equal(a["prior"].view(np.uint32), b["prior"].view(np.uint32))
        assert np.array_equal(ma, mb)
    s1 = sp1.engine.stats()
    assert s1["cache_hits"] > 0 and s1["cache_shared_rows"] > 0 and s1["error_flags"] == 0


def test_transpositions_within_one_board_hit_the_cache():
    """One board, a sharp evaluator (deep lines): positions reached by two move orders are evaluated once."""
    dev = torch.device("cuda", 0)
    sp0, ev0, t0 = _play(1, 600, 2, 0, dev, seed=5)
    sp1, ev1, t1 = _play(1, 600, 2, 16, dev, seed=5)
    for (a, ma), (b, mb) in zip(t0, t1):
        assert np.array_equal(a["visits"], b["visits"]) and np.array_equal(a["q"].view(np.uint32), b["q"].view(np.uint32)) and np.array_equal(ma, mb)
    s1 = sp1.engine.stats()
    assert s1["cache_shared_rows"] == 0 and s1["cache_hits"] > 0
    assert ev1.rows_asked == s1["cache_probes"] - s1["cache_hits"]


def test_slot_collisions_and_a_tiny_table_change_nothing():
    """2^10 entries for ~10^4 positions: entries are overwritten all the time, different keys meet on one slot within a step
    (each is evaluated on its own row, only the claim winner is stored) -- and the search is still the uncached search."""
    dev = torch.device("cuda", 0)
    B, n, moves = 32, 48, 5
    sp0, _, t0 = _play(B, n, moves, 0, dev, seed=9)
    sp1, _, t1 = _play(B, n, moves, 10, dev, seed=9)
    for (a, ma), (b, mb) in zip(t0, t1):
        assert np.array_equal(a["visits"], b["visits"]) and np.array_equal(a["q"].view(np.uint32), b["q"].view(np.uint32))
        assert np.array_equal(a["prior"].view(np.uint32), b["prior"].view(np.uint32)) and np.array_equal(ma, mb)
    s1 = sp1.engine.stats()
    assert s1["cache_stores"] > 1024          # the table was overwritten many times over


@pytest.mark.parametrize("B,layout", [(320, "auto"), (320, "g16"), (200, "g16")])
def test_real_net_planned_rows_and_cleared_on_new_weights(B, layout, monkeypatch):
    """The fused evaluator (k_pack_live_planes gather, stem and tower with device-side live-row counts, heads) on the planned
    rows: the same search as without a cache; a weight change empties the table. Board-major rows (what 320 boards get by
    default) and, forced, the group-of-16 row layout: 320 boards = 20 whole groups, 200 boards are padded to 13 groups."""
    monkeypatch.setenv("CCZ_CONV_LAYOUT", layout)
    from chinesechesszero_amd.net import PolicyValueNet
    from chinesechesszero_amd.selfplay import BatchedSelfPlay
    dev = torch.device("cuda", 0)
    n = 24
    res = []
    for log2 in (0, 15):
        torch.manual_seed(2)
        pvn = PolicyValueNet(device=dev, num_channels=256, resblocks_num=2)
        sp = BatchedSelfPlay(pvn.evaluate_leaves_logits, B, n_playout=n, seed=4, max_plies=30, eval_cache_log2=log2)
        assert sp.planned == (log2 > 0)
        rcs = []
        for _ in range(3):
            sp.run_move()
            rcs.append(sp.engine.root_children())
        res.append((sp, pvn, rcs))
    for a, b in zip(res[0][2], res[1][2]):
        assert np.array_equal(a["visits"], b["visits"]) and np.array_equal(a["acts"], b["acts"])
        assert np.array_equal(a["q"].view(np.uint32), b["q"].view(np.uint32)) and np.array_equal(a["prior"].view(np.uint32), b["prior"].view(np.uint32))
    sp, pvn, _ = res[1]
    s = sp.engine.stats()
    assert s["cache_shared_rows"] > B and s["cache_hits"] > 0
    # new weights: the cached evaluations are dropped before the next evaluation
    hits_before = s["cache_hits"]
    pvn.invalidate_inference_copy()
    sp.simulate()
    s2 = sp.engine.stats()
    assert s2["cache_hits"] == hits_before        # nothing could hit: the table was emptied
    sp.engine.check_healthy()
