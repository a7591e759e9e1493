"""GPU: the 'next' rows of SURVEY 8f -- batched evaluation matches and the UCI-style loop."""
import io

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_batched_match_plays_to_completion():
    from chinesechesszero_amd.match import BatchedMatch
    from chinesechesszero_amd.net import uniform_evaluator
    from test_gpu_soak import LinearEvaluator
    B = 48
    strong = LinearEvaluator(torch.device("cuda", 0), seed=1, sharp=8.0)
    m = BatchedMatch(strong, uniform_evaluator, B, n_playout=12, seed=5, max_plies=120)
    res = m.play()
    assert res["red_wins"] + res["black_wins"] + res["draws"] == B and res["unfinished"] == 0
    assert res["plies"].min() >= 1 and res["plies"].max() <= 120
    st = m.engine.stats()
    assert st["moves"] == int(res["plies"].sum())
    assert m.engine.game_status()["over"].all()


def test_uci_loop_go_returns_a_legal_move():
    from chinesechesszero_amd.game import Board
    from chinesechesszero_amd.uci import UciLoop
    from oracle.evaluators import hash_eval

    def policy(board, red_states=None, black_states=None):
        ids = board.legal_ids()
        p, v = hash_eval(board.squares()[None, :], np.array([1 if board.turn else 0]), salt=2, scale=40.0)
        return zip(ids, p[0][ids]), np.array([[v[0]]], dtype=np.float32)

    out = io.StringIO()
    loop = UciLoop(policy_value_fn=policy, n_playout=30, out=out)
    script = ["uci", "isready", "ucinewgame", "position startpos moves b2e2 h9g7", "go nodes 40", "position startpos moves a0a5", "d", "quit"]
    for line in script:
        if not loop.handle(line):
            break
    text = out.getvalue().splitlines()
    assert "uciok" in text and "readyok" in text
    best = [l for l in text if l.startswith("bestmove")]
    assert len(best) == 1
    b = Board()
    b.push("b2e2")
    b.push("h9g7")
    assert best[0].split()[1] in [m.uci() for m in b.legal_moves]
    assert any(l.startswith("info nodes 40") for l in text)
    assert any("illegal move a0a5" in l for l in text)  # rook cannot jump its own pawn... a0a5 is blocked by a3


def test_inference_net_fp16_fused_epilogue_matches_fp32_reference_architecture():
    """The evaluator's inference copy (BN folded, NHWC fp16, MIOpen convs + the one-pass HIP epilogue
    ccz_bias_act_f16) against the reference architecture in fp32 (tolerances of fp16 inference)."""
    from chinesechesszero_amd.net import InferenceNet, Net
    dev = torch.device("cuda", 0)
    torch.manual_seed(4)
    net = Net(64, 3).to(dev).eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 2)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
    x = torch.zeros(33, 17, 7, 10, 9, device=dev)
    x[:, 7] = (torch.rand(33, 7, 10, 9, device=dev) > 0.9).float()
    x[:, 15] = (torch.rand(33, 7, 10, 9, device=dev) > 0.9).float()
    x[::2, 16] = 1
    with torch.no_grad():
        logp, v = net(x)
        inf = InferenceNet(net).to(dev).eval()
        p16, v16 = inf(x.half())
        p32, v32 = InferenceNet(net, dtype=torch.float32).to(dev).eval()(x.half())
    assert torch.allclose(logp.exp(), p32, atol=1e-5) and torch.allclose(v.view(-1), v32, atol=1e-5)
    assert p16.dtype == torch.float32 and v16.dtype == torch.float32
    assert (logp.exp() - p16).abs().max().item() < 2e-3 and (v.view(-1) - v16).abs().max().item() < 2e-2
    assert torch.allclose(p16.sum(1), torch.ones(33, device=dev), atol=1e-3)
    # the epilogue kernel itself, bit-exact against the same fp16 op sequence in torch
    import ctypes as C
    from chinesechesszero_amd import _lib
    y = torch.randn(7, 64, 10, 9, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    r = torch.randn_like(y).contiguous(memory_format=torch.channels_last)
    b = torch.randn(64, device=dev, dtype=torch.float16)
    want1 = torch.relu(y + b.view(1, -1, 1, 1))
    want2 = torch.relu((y + b.view(1, -1, 1, 1)) + r)
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    y1, y2 = y.clone(memory_format=torch.preserve_format), y.clone(memory_format=torch.preserve_format)
    _lib.check(_lib.lib().ccz_bias_act_f16(s, C.c_void_p(y1.data_ptr()), C.c_void_p(b.data_ptr()), None, 7 * 90, 64))
    _lib.check(_lib.lib().ccz_bias_act_f16(s, C.c_void_p(y2.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(r.data_ptr()), 7 * 90, 64))
    assert torch.equal(y1, want1) and torch.equal(y2, want2)


def test_viewer_hook_shows_one_selected_board_of_the_batch_and_the_single_board_game():
    """SURVEY 8f row 3, second half: Game.graphic / BatchedSelfPlay.watch viewer hook-up (reference game.py:47-75, frontend.py:328)."""
    from examples.viewer import ChessWindow
    from chinesechesszero_amd.game import Board, Game
    from chinesechesszero_amd.mcts import MCTS_AI
    from chinesechesszero_amd.net import uniform_evaluator
    from chinesechesszero_amd.selfplay import BatchedSelfPlay
    w = ChessWindow("127.0.0.1", 0)          # not started: update_board only stores the state
    sp = BatchedSelfPlay(uniform_evaluator, 8, n_playout=6, seed=1)
    sp.watch(5, w)
    for ply in range(3):
        sp.run_move()
    assert w.updates == 3
    assert w._state["svg"].count("<circle") == int((sp.engine.root_positions()[5] != 0).sum())
    assert w._state["status"].startswith("board 5 - to move: black - ply: 3")
    assert 'fill-opacity="0.45"' in w._state["svg"]      # the last move is highlighted
    # the single-board loop: Game.graphic pushes every position when is_shown (game.py:103-104)
    orig = Board.is_game_over
    Board.is_game_over = lambda self: len(self.move_stack) >= 3 or orig(self)
    try:
        g = Game(viewer=w)
        a = MCTS_AI(uniform_evaluator, n_playout=5, is_selfplay=False)
        b = MCTS_AI(uniform_evaluator, n_playout=5, is_selfplay=False)
        g.start_play(a, b, is_shown=True)
    finally:
        Board.is_game_over = orig
    assert w.updates == 6 and "ply: 3" in w._state["status"]
