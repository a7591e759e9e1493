"""GPU: a board's games do not depend on how the job's boards are split over ranks (round 5: per-rank board counts).

RNG streams are keyed by the GLOBAL board id (Philox(seed, board id)), ``launch.board_partition`` hands every rank the id of its first
board as a prefix sum, and a board's evaluation is the same bits at every batch size -- so rank 0 carrying fewer boards than its peers
(the rank that shares its GPU with the trainer, BASELINE configs[4]) changes nothing about what any board plays."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _play(pvn, boards, base, moves, n):
    from chinesechesszero_amd.selfplay import BatchedSelfPlay
    sp = BatchedSelfPlay(pvn.evaluate_leaves_logits, boards, n_playout=n, seed=5, board_id_base=base, max_plies=40)
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
