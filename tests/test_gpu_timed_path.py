"""GPU: the path bench.py TIMES, at the BASELINE size, against the sequential oracle and against itself without the cache.

The timed region of the headline number is: cache probe -> plan -> pack of the live rows -> stem + 80 tower launches in the
group-of-16 row layout on two concurrent chains over the LIVE rows (device-side count) -> heads -> softmax + gather of the planned
rows + cache store -> fused k_step reading the engine-owned prior rows and leaf values. The reference's semantics for all of it is
one evaluation per playout (mcts.py:114). Two bit-exact checks at 4096 boards:

(a) a sample of the boards is mirrored on sequential oracles that are fed EXACTLY what the boundary hands the tree
    (``ccz_leaf_priors``: after table hits, shared rows and fresh evaluations have been merged) with the real 40 x 256 net:
    N / Q / P bit-exact across a move boundary;
(b) the same self-play with and without the evaluation cache (2-block 256-wide net: the same kernels and launch structure,
    40x cheaper): every root of every board identical over several moves with staggered game ends and restarts. This reaches
    k_cache_plan's four 1,024-board iterations, the padded last group and the live-range split of the concurrent chains.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def preroll(e, plies, mirror=None, stagger=True):
    """Diverge the boards the way bench.py's untimed setup does: ``plies`` lockstep plies, ONE stub-evaluator simulation each and
    a flat pi (temperature 1e3: a uniformly random legal move from the board's own Philox stream); with ``stagger`` board b is
    restarted at ply (b mod plies), so the boards end up on plies 1..plies of their games. ``mirror`` follows on its oracles."""
    from chinesechesszero_amd.net import uniform_evaluator
    B = e.B
    temps = np.full(B, 1e3, np.float64)
    b_idx = np.arange(B)
    for t in range(plies):
        if stagger and t > 0:
            mask = ((b_idx % plies) == t).astype(np.uint8)
            e.reset(mask)
            if mirror is not None:
                mirror.restarted(mask)
        leaf = e.select_leaves()
        e.expand_backup(*uniform_evaluator(leaf))
        moves = e.finish_move(temps=temps, keep_tree=False).cpu().numpy()
        if mirror is not None:
            mirror.played(moves, keep_tree=False)
        assert not e.game_status()["over"].any()
    e.check_healthy()


SAMPLE = [0, 1, 15, 16, 17, 1023, 1024, 2047, 2048, 3071, 3072, 4079, 4080, 4094, 4095]


def test_planned_boundary_real_net_4096_boards_vs_sequential_oracle():
    """(a): 4096 boards, the real 40 x 256 net, a 2^20-entry cache, 2 moves x 40 simulations = 80 simulations over a move
    boundary, 32 sampled boards (group borders, chain borders, the last group + random ones)."""
    from gpu_harness import SampleMirror
    from chinesechesszero_amd.net import PolicyValueNet
    from chinesechesszero_amd.selfplay import BatchedSelfPlay
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    pvn = PolicyValueNet(device=dev)                      # 40 x 256, random init: what bench.py runs
    B, n = 4096, 40
    sp = BatchedSelfPlay(pvn.evaluate_leaves_logits, B, n_playout=n, seed=7, max_plies=64, eval_cache_log2=20)
    assert sp.planned
    e = sp.engine
    rs = np.random.RandomState(1)
    sample = sorted(set(SAMPLE) | set(rs.choice(B, size=17, replace=False).tolist()))[:32]
    sm = SampleMirror(e, sample, check_every=4)
    preroll(e, 6, mirror=sm)                               # diverged positions, boards on plies 1..6 of their games
    rows_seen = []
    for move in range(2):
        leaf = e.select_leaves()
        for i in range(n):
            lg, v = sp._planned_eval(leaf)                 # probe + plan + the network on the planned rows (g16, two chains)
            rows_seen.append(int(e.n_miss.item()))
            e.gather_priors_planned(lg, v)                 # softmax + gather of the planned rows, cache store
            pri, val = e.leaf_priors()                     # what the tree is about to consume
            sm.backup_on_oracles_compact(pri, val)
            if i + 1 < n:
                leaf = e.step_compact(None)
            else:
                e.expand_backup_compact(None)
        rc = sm.compare_roots()                            # N, Q, P of the sampled boards: bit-exact vs the sequential oracle
        assert np.all(rc["root_visits"][sample] >= n)
        moves = sp.finish_move().cpu().numpy()
        sm.played(moves)
        sm.compare_roots()                                 # the kept subtrees == the oracle's update_with_move
    e.check_healthy()
    st = e.stats()
    assert st["sims"] >= 2 * B * n and st["error_flags"] == 0
    assert st["cache_hits"] > 0 and st["cache_shared_rows"] > 0 and st["cache_stores"] > 0
    # the evaluator really ran on a compacted batch that spans several 1,024-board plan iterations and both chains
    assert max(rows_seen) > 3072 and min(rows_seen) < B
    assert pvn._infer._g16(B)                              # the group-of-16 tower path (auto layout at this size)
    e.close()


def _selfplay_trace(cache_log2, B=4096, n=24, moves=4, verify=False):
    from chinesechesszero_amd.net import PolicyValueNet
    from chinesechesszero_amd.selfplay import BatchedSelfPlay
    dev = torch.device("cuda", 0)
    torch.manual_seed(2)
    pvn = PolicyValueNet(device=dev, num_channels=256, resblocks_num=2)
    sp = BatchedSelfPlay(pvn.evaluate_leaves_logits, B, n_playout=n, seed=4, max_plies=6, eval_cache_log2=cache_log2,
                         cache_verify=verify)
    assert sp.planned == (cache_log2 > 0)
    e = sp.engine
    preroll(e, 4)                                          # boards on plies 1..4: the 6-ply cap ends games at moves 2, 3, 4, ...
    trace, restarts = [], 0
    for _ in range(moves):
        sp.search()
        rc = e.root_children()
        mv = sp.finish_move().cpu().numpy().copy()
        st = e.game_status()
        trace.append((rc, mv, st["over"].copy(), st["plies"].copy()))
        if st["over"].any():
            restarts += int(st["over"].sum())
            for _chunk in e.harvest_chunks(1 << 16):
                pass
    e.check_healthy()
    stats = e.stats()
    g16 = pvn._infer._g16(B)
    e.close()
    return trace, stats, restarts, g16


def test_cache_on_vs_off_bit_identical_at_4096_boards_with_restarts():
    """(b): all 4096 roots -- ids, visit counts, Q, priors --, the moves played and the game ends, cache on vs cache off."""
    t0, s0, r0, g0 = _selfplay_trace(0)
    t1, s1, r1, g1 = _selfplay_trace(20)
    assert g0 and g1                                       # both ran the group-of-16 tower kernels
    assert r0 == r1 and r0 > 1024                          # staggered game ends: boards restarted inside the compared window
    for (a, ma, oa, pa), (b, mb, ob, pb) in zip(t0, t1):
        for key in ("k", "acts", "visits", "root_visits"):
            assert np.array_equal(a[key], b[key]), key
        assert np.array_equal(a["q"].view(np.uint32), b["q"].view(np.uint32))
        assert np.array_equal(a["prior"].view(np.uint32), b["prior"].view(np.uint32))
        assert np.array_equal(ma, mb) and np.array_equal(oa, ob) and np.array_equal(pa, pb)
    for key in ("sims", "moves", "games", "expansions", "terminal_leaves", "sum_depth", "sum_children", "nodes_peak", "depth_peak"):
        assert s0[key] == s1[key], key
    assert s0["cache_probes"] == 0 and s1["cache_probes"] == s1["expansions"] - 4 * 4096   # (the preroll's 4 dense stub plies)
    computed = s1["cache_probes"] - s1["cache_hits"] - s1["cache_shared_rows"]
    assert 0 < computed < s1["cache_probes"] and s1["cache_hits"] > 0 and s1["cache_shared_rows"] > 0


def test_cache_verify_mode_counts_no_mismatch_and_changes_nothing():
    """CCZ_FLAG_CACHE_VERIFY: ~1 hit in 128 goes through the evaluator again and must come back bit-identical; the search is the
    search without the flag."""
    t1, s1, _, _ = _selfplay_trace(20, B=1024, n=32, moves=4)
    t2, s2, _, _ = _selfplay_trace(20, B=1024, n=32, moves=4, verify=True)
    for (a, ma, _, _), (b, mb, _, _) in zip(t1, t2):
        assert np.array_equal(a["visits"], b["visits"]) and np.array_equal(a["q"].view(np.uint32), b["q"].view(np.uint32))
        assert np.array_equal(a["prior"].view(np.uint32), b["prior"].view(np.uint32)) and np.array_equal(ma, mb)
    assert s1["cache_verified"] == 0 and s1["cache_verify_mismatches"] == 0
    assert s2["cache_verified"] > 0 and s2["cache_verify_mismatches"] == 0
    assert s2["cache_hits"] + s2["cache_verified"] == s1["cache_hits"]       # a verified hit is a hit that was computed again
    assert 0.5 / 128 < s2["cache_verified"] / s1["cache_hits"] < 2.0 / 128


def test_cache_verify_mode_catches_stale_entries():
    """The negative control: weights change and the table is NOT cleared (the engine is driven below BatchedSelfPlay's version
    check): verified hits now disagree and are counted."""
    from chinesechesszero_amd.net import PolicyValueNet
    from chinesechesszero_amd.selfplay import BatchedSelfPlay
    dev = torch.device("cuda", 0)
    torch.manual_seed(2)
    pvn = PolicyValueNet(device=dev, num_channels=256, resblocks_num=2)
    B, n = 1024, 32
    sp = BatchedSelfPlay(pvn.evaluate_leaves_logits, B, n_playout=n, seed=4, max_plies=30, eval_cache_log2=18, cache_verify=True)
    e = sp.engine
    sp.run_move()
    e.reset()                                            # the same openings again: ~30 k hits, ~1 % of them evaluated again
    sp.run_move()
    s = e.stats()
    assert s["cache_hits"] > 10000 and s["cache_verified"] > 50 and s["cache_verify_mismatches"] == 0
    with torch.no_grad():
        for p in pvn.policy_value_net.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    pvn.refresh_inference_copy()
    sp._cache_version = sp._evaluator_version()          # suppress the invalidation BatchedSelfPlay would do
    e.reset()                                            # the same openings again: plenty of hits on stale entries
    sp.run_move()
    s2 = e.stats()
    assert s2["cache_verified"] > s["cache_verified"] and s2["cache_verify_mismatches"] > 0
    e.close()
