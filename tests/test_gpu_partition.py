"""GPU: a board's games do not depend on how the job's boards are split over ranks (round 5: per-rank board counts).

RNG streams are keyed by the GLOBAL board id (Philox(seed, board id)), ``launch.board_partition`` hands every rank the id of its first
board as a prefix sum, and a board's evaluation is the same bits at every batch size -- so rank 0 carrying fewer boards than its peers
(the rank that shares its GPU with the trainer, BASELINE configs[4]) changes nothing about what any board plays."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _play(pvn, boards, base, moves, n, cache_log2=0):
    from chinesechesszero_amd.selfplay import BatchedSelfPlay
    sp = BatchedSelfPlay(pvn.evaluate_leaves_logits, boards, n_playout=n, seed=5, board_id_base=base, max_plies=40, eval_cache_log2=cache_log2)
    out = []
    for _ in range(moves):
        sp.search()
        rc = sp.engine.root_children()
        mv = sp.finish_move().cpu().numpy()
        out.append((mv.copy(), rc["k"].copy(), rc["acts"].copy(), rc["visits"].copy(), rc["q"].copy().view(np.uint32)))
    sp.engine.check_healthy()
    return out


def test_unequal_board_counts_play_the_same_games_as_equal_ones():
    from chinesechesszero_amd.launch import board_partition
    from chinesechesszero_amd.net import PolicyValueNet
    torch.manual_seed(3)
    pvn = PolicyValueNet(device="cuda:0", num_channels=256, resblocks_num=1)
    pvn.refresh_inference_copy()
    moves, n = 5, 24
    whole = _play(pvn, 16, 0, moves, n)                       # one rank holding all 16 boards
    for boards_rank0 in (8, 4):                               # two ranks: equal halves; a light rank 0 next to a full peer
        counts, bases = board_partition(2, 16 - boards_rank0, boards_rank0)
        assert sum(counts) == 16 and bases == [0, boards_rank0]
        parts = [_play(pvn, c, b, moves, n) for c, b in zip(counts, bases)]
        for t in range(moves):
            for field in range(5):
                joined = np.concatenate([p[t][field] for p in parts])
                assert np.array_equal(joined, whole[t][field]), (boards_rank0, t, field)
    # and the streams do differ between boards (the test is not comparing constants)
    assert len({tuple(int(m[0][b]) for m in whole) for b in range(16)}) > 4


def test_the_evaluation_cache_does_not_tie_a_board_to_its_rank_mates():
    """The same with the planned evaluator boundary: 256 boards on one rank share ONE evaluation cache (a position evaluated for one
    board serves every other), split as 64 + 192 they share two smaller ones -- and every board still plays the same games, because a
    cached evaluation is the evaluation (bit for bit) and the streams follow the global board id."""
    from chinesechesszero_amd.net import PolicyValueNet
    torch.manual_seed(7)
    pvn = PolicyValueNet(device="cuda:0", num_channels=256, resblocks_num=1)
    pvn.refresh_inference_copy()
    moves, n = 3, 16
    whole = _play(pvn, 256, 0, moves, n, cache_log2=16)
    parts = [_play(pvn, 64, 0, moves, n, cache_log2=16), _play(pvn, 192, 64, moves, n, cache_log2=16)]
    plain = _play(pvn, 256, 0, moves, n, cache_log2=0)
    for t in range(moves):
        for field in range(5):
            joined = np.concatenate([p[t][field] for p in parts])
            assert np.array_equal(joined, whole[t][field]), (t, field)
            assert np.array_equal(plain[t][field], whole[t][field]), (t, field)      # and the cache changed nothing in the first place
