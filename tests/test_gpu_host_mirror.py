"""GPU: the host mirror of the reference call surface (tools / mcts / game / collect) on the HIP engine."""
import os

import numpy as np
import torch
import pytest

pytestmark = pytest.mark.gpu


def _hash_policy(salt=7, scale=40.0):
    from oracle.evaluators import hash_eval

    def policy(board, red_states=None, black_states=None):
        ids = board.legal_ids()
        p, v = hash_eval(board.squares()[None, :], np.array([1 if board.turn else 0]), salt=salt, scale=scale)
        return zip(ids, p[0][ids]), np.array([[v[0]]], dtype=np.float32)

    return policy


def test_mcts_ai_get_action_reproduces_reference_under_seed(golden):
    """MCTS_AI.get_action (self-play, tree reuse) == the reference's own get_action for 3 plies, same np.random.seed."""
    from chinesechesszero_amd.game import Board, Move
    from chinesechesszero_amd.mcts import MCTS_AI
    d = golden["data"]
    ai = MCTS_AI(_hash_policy(), c_puct=5, n_playout=120, is_selfplay=True)
    board = Board()
    np.random.seed(2024)
    calls = []
    for ply in range(3):
        move, move_probs = ai.get_action(board, temp=1.0, return_prob=True, on_playout=lambda n: calls.append(n))
        assert int(move) == int(d[f"getaction_p{ply}_move"])
        assert np.array_equal(move_probs, d[f"getaction_p{ply}_probs"])
        board.push(Move.from_id(move))
    ref_calls = golden["meta"]["on_playout_calls"]
    assert calls[:len(ref_calls)] == ref_calls and sum(calls) == 360


def test_board_rules_match_oracle_along_random_games():
    from oracle import OracleBoard
    from chinesechesszero_amd.game import Board, Move
    rs = np.random.RandomState(5)
    for g in range(3):
        b, o = Board(), OracleBoard()
        for ply in range(80):
            ids = o.legal_ids()
            assert b.legal_ids() == ids
            assert [m.uci() for m in b.legal_moves] == o.legal_moves
            assert b.is_game_over() == o.is_game_over()
            assert b.is_check() == o.in_check()
            assert b.is_fourfold_repetition() == o.is_fourfold_repetition()
            if not ids or o.is_game_over():
                break
            m = ids[rs.randint(len(ids))]
            b.push(Move.from_id(m))
            o.push_id(m)
            assert np.array_equal(b.squares(), o.squares()) and b.turn == o.turn and b.halfmove_clock == o.halfmove
    # repetition through the host bookkeeping
    b = Board()
    for _ in range(3):
        for u in ("b0c2", "b9c7", "c2b0", "c7b9"):
            b.push(u)
    assert b.is_fourfold_repetition() and b.is_game_over() and b.outcome().winner is None
    assert b.fen().startswith("rnbakabnr/9/1c5c1/p1p1p1p1p/9/9/P1P1P1P1P/1C5C1/9/RNBAKABNR w")


def test_match_play_discards_tree_and_syncs_opponent_moves():
    from chinesechesszero_amd.game import Board, Move
    from chinesechesszero_amd.mcts import MCTS_AI
    ai = MCTS_AI(_hash_policy(salt=3), c_puct=5, n_playout=40, is_selfplay=False)
    board = Board()
    np.random.seed(1)
    for ply in range(4):
        move = ai.get_action(board)       # temp 1e-3: (almost) argmax of visits
        rc = ai.mcts.root_children()
        assert rc["root_visits"] == 40    # fresh tree every move (mcts.py:228-229)
        chosen = rc["visits"][rc["acts"].tolist().index(int(move))]
        assert chosen == rc["visits"].max()   # ties between equally visited children are split by the sampler
        board.push(Move.from_id(move))
        reply = board.legal_ids()[0]      # opponent's move the search never saw
        board.push(Move.from_id(reply))


def test_game_start_self_play_tuples():
    """One short self-play game through Game/MCTS_AI; tuples follow game.py:195-237 (fixed history by default)."""
    from chinesechesszero_amd.game import Game
    from chinesechesszero_amd.mcts import MCTS_AI
    from chinesechesszero_amd.tools import decode_board
    from chinesechesszero_amd.game import Board

    class ShortGame(Game):
        """stop after 6 plies by declaring the game over (keeps the test short)"""

    ai = MCTS_AI(_hash_policy(salt=1), c_puct=5, n_playout=16, is_selfplay=True)
    np.random.seed(3)
    g = Game()
    # monkeypatch Board.is_game_over to end after 6 plies
    orig = Board.is_game_over
    Board.is_game_over = lambda self: len(self.move_stack) >= 6 or orig(self)
    try:
        data = g.start_self_play(ai)
    finally:
        Board.is_game_over = orig
    assert len(data) == 6
    start_red, start_black = decode_board(Board())
    red0, black0, pi0, z0 = data[0]
    assert len(red0) == 8 and np.array_equal(red0[0], start_red) and np.array_equal(red0[7], start_red)
    assert abs(pi0.sum() - 1.0) < 1e-12 and pi0.shape == (2086,)
    red5 = data[5][0]
    assert not np.array_equal(red5[0], red5[5]) or not np.array_equal(data[5][1][0], data[5][1][5])
    assert all(z == 0 for *_, z in data)  # no outcome: draw-valued tuples


def test_collect_pipeline_batched_writes_trainer_format(tmp_path):
    from chinesechesszero_amd.collect import CollectPipeline
    cp = CollectPipeline(init_model=None, n_boards=16, n_playout=4, data_dir=str(tmp_path), num_channels=16, resblocks_num=1)
    cp.load_model()
    from chinesechesszero_amd.selfplay import BatchedSelfPlay
    cp.selfplay = BatchedSelfPlay(cp.policy_value_net.evaluate_leaves, 16, n_playout=4, max_plies=5, seed=2)
    cp.collect_batched(7)
    assert not (tmp_path / "states.npy").exists() and cp.sink.rows() == 16 * 5 * 2   # rows stay in shards while collecting
    assert cp.sink.finalize() == 16 * 5 * 2                                          # the converter step (convert.py:21-107)
    import json
    meta = json.load(open(tmp_path / "meta.json"))
    assert list(meta)[:7] == ["total_count", "states_shape", "states_dtype", "mcts_shape", "mcts_dtype", "winners_shape", "winners_dtype"]
    assert meta["total_count"] == 160 and meta["mcts_dtype"] == "float64" and meta["iters"] == 16
    states = np.load(tmp_path / "states.npy")
    pi = np.load(tmp_path / "mcts.npy")
    z = np.load(tmp_path / "winners.npy")
    assert states.dtype == np.float16 and states.shape[1:] == (17, 7, 10, 9)
    assert pi.dtype == np.float64 and z.dtype == np.float32                           # the dtypes the reference stores
    assert pi.shape == (states.shape[0], 2086) and z.shape == (states.shape[0],)
    assert states.shape[0] == 16 * 5 * 2  # every board was adjudicated at 5 plies once, mirrored
    assert np.allclose(pi.sum(1), 1.0, atol=1e-4)
    assert cp.iters == 16


def test_record_shards_and_dense_shards_give_the_same_trainer_files(tmp_path):
    """Round 5: the batched collector writes compact ply records while it runs (12.5 MB per move of 4096 boards instead of 1.07 GB of
    dense rows) and TupleSink.finalize() expands them on the GPU: the trainer's files are byte for byte those of the dense path, also
    in the reference's quirk mode and when the expansion is redone after an interruption."""
    from chinesechesszero_amd.collect import CollectPipeline, TupleSink
    from chinesechesszero_amd.selfplay import BatchedSelfPlay
    for quirks in (False, True):
        out = {}
        for name in ("dense", "records"):
            d = tmp_path / f"{name}{int(quirks)}"
            cp = CollectPipeline(init_model=None, n_boards=16, n_playout=4, data_dir=str(d), num_channels=16, resblocks_num=1,
                                 dense_shards=(name == "dense"), reference_quirks=quirks)
            torch.manual_seed(4)
            cp.load_model()
            cp.selfplay = BatchedSelfPlay(cp.policy_value_net.evaluate_leaves, 16, n_playout=4, max_plies=5, seed=2, reference_quirks=quirks)
            cp.collect_batched(13)
            assert cp.sink.rows() == 16 * 5 * 2 * 2 and cp.iters == 32
            shard_bytes = sum(os.path.getsize(d / f) for f in os.listdir(d) if f.startswith((".shard_", ".rshard_")))
            if name == "records":
                assert shard_bytes < 16 * 5 * 2 * 1000                         # 880 B per ply (+ .npy headers)
                # an expansion that was interrupted after its first dense shard: the next sink starts it again, nothing counts twice
                cp.sink.close()
                rs = sorted(f for f in os.listdir(d) if f.startswith(".rshard_"))
                tag = rs[0][len(".rshard_"):-len(".npy")]
                for sfx, arr in (("_s.npy", np.zeros((3, 17, 7, 10, 9), np.float16)), ("_p.npy", np.zeros((3, 2086), np.float64)), ("_z.npy", np.zeros(3, np.float32))):
                    np.save(d / f".shard_r{tag}_0000{sfx}", arr)
                cp.sink = TupleSink(str(d))
            else:
                assert shard_bytes > 16 * 5 * 2 * 2 * 29768
            assert cp.sink.finalize() == 320
            assert not [f for f in os.listdir(d) if f.startswith((".shard_", ".rshard_"))]
            out[name] = [np.load(d / f) for f in ("states.npy", "mcts.npy", "winners.npy")]
        for a, b in zip(out["dense"], out["records"]):
            assert a.dtype == b.dtype and np.array_equal(a, b)


def test_collect_pipeline_batched_through_the_async_exchange_stores_the_same_rows(tmp_path):
    """collect_batched with replay.AsyncRecordExchange (a group of one here: the backlog is handed straight over): finished games
    leave as compact records, are expanded and stored -- the same files, byte for byte, as the direct dense harvest."""
    from chinesechesszero_amd.collect import CollectPipeline
    from chinesechesszero_amd.replay import AsyncRecordExchange
    from chinesechesszero_amd.selfplay import BatchedSelfPlay
    out = {}
    for name in ("direct", "exchange"):
        d = tmp_path / name
        cp = CollectPipeline(init_model=None, n_boards=16, n_playout=4, data_dir=str(d), num_channels=16, resblocks_num=1)
        torch.manual_seed(4)
        cp.load_model()
        cp.selfplay = BatchedSelfPlay(cp.policy_value_net.evaluate_leaves, 16, n_playout=4, max_plies=5, seed=2)
        ex = AsyncRecordExchange(64, "cuda:0") if name == "exchange" else None
        cp.collect_batched(13, gatherer=ex)
        if ex is not None:
            assert cp.drain_exchange(ex) == 32 and ex.issued == ex.completed >= 2
        assert cp.sink.finalize() == 16 * 5 * 2 * 2 and cp.iters == 32           # two adjudicated games per board
        out[name] = [np.load(d / f) for f in ("states.npy", "mcts.npy", "winners.npy")]
    for a, b in zip(out["direct"], out["exchange"]):
        assert a.dtype == b.dtype and np.array_equal(a, b)


def test_start_play_two_players_two_engines():
    """game.py:77-130 with two MCTS_AI players (each owns an engine): moves alternate, both stay in sync."""
    from chinesechesszero_amd.game import Board, Game
    from chinesechesszero_amd.mcts import MCTS_AI
    red = MCTS_AI(_hash_policy(salt=11), c_puct=5, n_playout=24, is_selfplay=False)
    black = MCTS_AI(_hash_policy(salt=12), c_puct=5, n_playout=24, is_selfplay=False)
    np.random.seed(9)
    orig = Board.is_game_over
    Board.is_game_over = lambda self: len(self.move_stack) >= 7 or orig(self)
    try:
        g = Game()
        winner = g.start_play(red, black, is_shown=False)
    finally:
        Board.is_game_over = orig
    assert len(g.board.move_stack) == 7 and winner == -1  # cut short: no outcome -> reported as a draw
    assert red.player == 1 and black.player == 0
    # each engine followed the whole game (its own moves and the opponent's replies)
    assert np.array_equal(red.mcts._engine.root_positions()[0], Board(g.board.squares(), g.board.turn).squares()) or len(red.mcts._synced) >= 5
    assert black.mcts._synced == [m.id for m in g.board.move_stack][:len(black.mcts._synced)]


def test_engine_create_destroy_does_not_leak_device_memory():
    """Six engines created and destroyed must leave the device's free memory where it was. ``mem_get_info`` is the DRIVER's device-wide
    figure and hipFree returns before the kernel driver has released the buffers (seen once in eight whole-suite runs: all six engines'
    1.16 GB each still counted right after the loop), so the figure is polled for a while, and a pass that still reads high is repeated
    once from a fresh baseline: a leak grows again, a late release does not."""
    import time
    import torch
    from chinesechesszero_amd.engine import SelfPlayEngine

    def one_pass():
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        for _ in range(6):
            e = SelfPlayEngine(256, n_playout=100)
            e.select_leaves()
            torch.cuda.synchronize()
            e.close()
            del e
        deadline = time.time() + 15.0
        while True:
            torch.cuda.empty_cache()
            torch.cuda.synchronize()
            free1, _ = torch.cuda.mem_get_info()
            if free0 - free1 < 64 * 1024 * 1024 or time.time() > deadline:
                return free0, free1
            time.sleep(0.25)

    free0, free1 = one_pass()
    if free0 - free1 >= 64 * 1024 * 1024:
        free0, free1 = one_pass()
    assert free0 - free1 < 64 * 1024 * 1024, (free0, free1)


def test_collect_pipeline_single_board_host_path(tmp_path):
    """collect.py:133-176 in its own control flow: one self-play game through Game/MCTS_AI on a B=1 engine,
    preprocess + flip_data on the host, rows appended in the trainer's format."""
    from chinesechesszero_amd.collect import CollectPipeline
    from chinesechesszero_amd.game import Board
    from chinesechesszero_amd.tools import flip_map
    cp = CollectPipeline(init_model=None, n_boards=1, n_playout=12, data_dir=str(tmp_path), num_channels=16, resblocks_num=1)
    orig = Board.is_game_over
    Board.is_game_over = lambda self: len(self.move_stack) >= 5 or orig(self)
    np.random.seed(4)
    try:
        iters = cp.collect_data()
    finally:
        Board.is_game_over = orig
    assert iters == 1 and cp.episode_len == 5
    cp.sink.finalize()
    states = np.load(tmp_path / "states.npy")
    pi = np.load(tmp_path / "mcts.npy")
    z = np.load(tmp_path / "winners.npy")
    assert states.shape == (10, 17, 7, 10, 9) and pi.shape == (10, 2086) and z.shape == (10,)
    fm = flip_map()
    for t in range(5):
        assert np.array_equal(states[5 + t], states[t][:, :, :, ::-1])       # np.flip(axis=2) of every group
        assert np.allclose(pi[5 + t], pi[t][fm]) and z[5 + t] == z[t]
        assert np.all(states[t][16] == (1 if t % 2 == 0 else 0))               # side-to-move plane (fixed mode)
    assert np.allclose(pi.sum(1), 1.0, atol=1e-6)


def test_policy_value_fn_matches_the_reference_end_to_end():
    """PolicyValueNet.policy_value_fn (net.py:151-205) against the reference's own function run on the same positions with
    the same closed-form weights (tests/golden/make_golden_net.py): legal ids in the same order, exp(log p) of exactly those
    ids, value as an ndarray (1,1) float32 -- the net on the CPU path (float32, as the golden), the rules on the GPU."""
    import json
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import net_recipe
    from chinesechesszero_amd.game import Board
    from chinesechesszero_amd.net import PolicyValueNet
    d = dict(np.load(os.path.join(here, "golden", "reference_net.npz")))
    meta = json.load(open(os.path.join(here, "golden", "reference_net.json")))
    pvn = PolicyValueNet(use_gpu=False, device="cpu")
    net_recipe.fill_state_dict(pvn.policy_value_net)
    for name in ("start", "wide80_black"):
        board = Board(d[f"pvfn_{name}_sq"], bool(d[f"pvfn_{name}_turn"]))
        act_probs, value = pvn.policy_value_fn(board)
        pairs = list(act_probs)
        assert [a for a, _ in pairs] == d[f"pvfn_{name}_ids"].tolist()
        assert np.allclose(np.array([p for _, p in pairs], np.float32), d[f"pvfn_{name}_probs"], rtol=2e-3, atol=1e-8)
        value = np.asarray(value)
        assert list(value.shape) == meta[f"pvfn_{name}_value_shape"] == [1, 1] and str(value.dtype) == meta[f"pvfn_{name}_value_dtype"] == "float32"
        assert np.allclose(value, d[f"pvfn_{name}_value"], rtol=0, atol=5e-4)
