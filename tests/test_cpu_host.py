"""CPU: host-side logic and the C-ABI library surface (no GPU compute)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    """Every function include/cczero.h declares is exported by libcczero.so and bound by the shim."""
    from chinesechesszero_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "cczero.h")).read()
    declared = set(re.findall(r"\b(ccz_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ccz_engine", "ccz_config", "ccz_stats"}
    assert len(declared) >= 20
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), name
        assert name in _lib.PROTOTYPES, f"{name} not bound in _lib.PROTOTYPES"
    assert set(_lib.PROTOTYPES) == declared
    assert L.ccz_abi_version() == 7 == _lib.ABI_VERSION   # ABI 7: CCZ_FLAG_STRICT + CCZ_ERR_PRUNED / _TRUNCATED, CCZ_CONV_G16_PERSISTENT; ABI 5: ccz_leaf_priors, CCZ_FLAG_CACHE_VERIFY + two ccz_stats counters; ABI 6: ccz_conv3x3_c256_heads_f16, ccz_fc_f16 relu bits 1 / 2
    assert ctypes.sizeof(_lib.Config) == 96 and ctypes.sizeof(_lib.Stats) == 152      # ABI 3: four evaluation-cache counters appended; ABI 5: two verify counters
    # ABI 2 fields sit where include/cczero.h puts them (pointer at 64, plane map at 72, rule flags at 80); ABI 3 gives the word
    # behind rule_flags a meaning (eval_cache_log2) without moving anything
    assert _lib.Config.move_rank_host.offset == 64 and _lib.Config.plane_of_type.offset == 72 and _lib.Config.rule_flags.offset == 80 and _lib.Config.type_rank.offset == 88
    assert _lib.Config.eval_cache_log2.offset == 84 and _lib.Stats.cache_probes.offset == 104
    m = re.search(r"#define CCZ_ABI_VERSION (\d+)", hdr)
    assert m and int(m.group(1)) == 7


def test_tables_from_library_match_reference_golden(golden):
    from chinesechesszero_amd import tools
    assert [tools.move_id2move_action[i] for i in range(2086)] == golden["table"]
    assert all(tools.move_action2move_id[s] == i for i, s in enumerate(golden["table"]))
    assert np.array_equal(tools.flip_map(), golden["data"]["flip_map"])
    for i in (0, 17, 2037, 2038, 2085):
        s = golden["table"][i]
        assert tools.move_action2move_id[tools.flip(s)] == golden["data"]["flip_map"][i]
    assert tools.flip("d9e8") == "f9e8"
    x = np.array([0.5, 2.0, -1.0])
    assert np.allclose(tools.softmax(x), np.exp(x) / np.exp(x).sum())


def test_engine_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from chinesechesszero_amd._lib import CczError
    from chinesechesszero_amd.engine import SelfPlayEngine, legal_moves
    with pytest.raises(CczError):
        SelfPlayEngine(4)
    with pytest.raises(CczError):
        legal_moves(np.zeros((1, 90), np.uint8), np.zeros(1, np.uint8))
    # the C ABI itself also refuses: no CPU fallback behind the boundary
    from chinesechesszero_amd import _lib
    L = _lib.lib()
    cfg = _lib.Config(n_boards=2, n_playout=4, c_puct=5, eps=0.25, alpha=0.2, temp=1.0)
    h = ctypes.c_void_p()
    assert L.ccz_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b"no HIP device" in L.ccz_last_error() or b"failed" in L.ccz_last_error()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "chinesechesszero_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f


def test_inference_copy_equals_reference_architecture():
    from chinesechesszero_amd.net import InferenceNet, Net
    torch.manual_seed(1)
    net = Net(16, 2).eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 2)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
    assert sorted(net.state_dict().keys())[:3] == ["conv_block.bias", "conv_block.weight", "conv_block_bn.bias"]
    assert "res_blocks.1.conv2_bn.running_var" in net.state_dict() and "value_fc2.weight" in net.state_dict()
    x = torch.zeros(6, 17, 7, 10, 9)
    x[:, 7] = (torch.rand(6, 7, 10, 9) > 0.9).float()
    x[:, 15] = (torch.rand(6, 7, 10, 9) > 0.9).float()
    x[::2, 16] = 1
    with torch.no_grad():
        logp, v = net(x)
        for live in (True, False):
            p2, v2 = InferenceNet(net, dtype=torch.float32, live_only=live)(x.half())
            assert torch.allclose(logp.exp(), p2, atol=1e-6) and torch.allclose(v.view(-1), v2, atol=1e-6)
    full = Net()
    assert sum(p.numel() for p in full.parameters()) == 50883979  # SURVEY: reference net size


def test_tuple_sink_and_replay_buffer(tmp_path):
    from chinesechesszero_amd.collect import TupleSink
    from chinesechesszero_amd.replay import ReplayBuffer
    sink = TupleSink(str(tmp_path))
    s = torch.zeros(5, 17, 7, 10, 9, dtype=torch.float16)
    s[:, 16] = 1
    p = torch.rand(5, 2086)
    z = torch.tensor([1., -1, 0, 1, -1])
    sink.append(s, p, z, games=1)
    assert sink.flush() == 5
    sink2 = TupleSink(str(tmp_path))
    sink2.append(s[:2], p[:2], z[:2], games=1)
    assert sink2.flush() == 7
    st = np.load(tmp_path / "states.npy", mmap_mode="r")
    assert st.shape == (7, 17, 7, 10, 9) and st.dtype == np.float16
    assert np.load(tmp_path / "winners.npy").tolist() == [1, -1, 0, 1, -1, 1, -1]
    import json
    assert json.load(open(tmp_path / "meta.json"))["iters"] == 2
    rb = ReplayBuffer(8, "cpu")
    rb.append(s, p, z)
    rb.append(s, p, z)
    assert rb.size == 8 and rb.total == 10 and rb.head == 2
    a, b, c = rb.sample(4)
    assert a.shape == (4, 17, 7, 10, 9) and b.shape == (4, 2086) and c.shape == (4,)


def test_uci_fen_parsing_roundtrip():
    from chinesechesszero_amd.game import Board, start_squares
    from chinesechesszero_amd.uci import board_from_fen, parse_position
    b = parse_position(["startpos"])
    assert np.array_equal(b.squares(), start_squares()) and b.turn is True
    fen = Board().fen()
    assert fen == "rnbakabnr/9/1c5c1/p1p1p1p1p/9/9/P1P1P1P1P/1C5C1/9/RNBAKABNR w - - 0 1"
    b2 = board_from_fen(fen)
    assert np.array_equal(b2.squares(), start_squares()) and b2.turn is True and b2.halfmove_clock == 0
    b3 = board_from_fen("4k4/9/9/9/9/9/9/9/4R4/3K5 b - - 37 60")
    assert b3.turn is False and b3.halfmove_clock == 37 and b3.piece_at(3).piece_type == 7 and b3.piece_at(13).piece_type == 3
    assert b3.piece_at(85).color is False and b3.fen().startswith("4k4/9/9/9/9/9/9/9/4R4/3K5 b")
    with pytest.raises(ValueError):
        board_from_fen("9/9/9 w")
    with pytest.raises(ValueError):
        parse_position(["fen"])
    # host-side push bookkeeping (no rules involved)
    b.push("b2e2")
    assert b.turn is False and b.halfmove_clock == 1 and b.piece_at(4 + 18).piece_type == 2 and b.peek().uci() == "b2e2"
    b.push("h9g7")
    b.push("e2e6")  # cannon takes the pawn: clock resets
    assert b.halfmove_clock == 0 and len(b.move_stack) == 3 and len(b._chain) == 1


def test_trainer_step_reduces_loss_on_fixed_batch():
    """The config-5 consumer: the update of train.py:163-187 on a fixed synthetic batch (CPU, fp32)."""
    from chinesechesszero_amd.net import PolicyValueNet
    from chinesechesszero_amd.trainer import Trainer
    torch.manual_seed(0)
    pvn = PolicyValueNet(use_gpu=False, device="cpu", num_channels=16, resblocks_num=1)
    tr = Trainer(pvn, lr=2e-3)
    g = torch.Generator().manual_seed(1)
    states = (torch.rand((32, 17, 7, 10, 9), generator=g) > 0.9).half()
    pi = torch.zeros(32, 2086)
    pi[torch.arange(32), torch.randint(0, 2086, (32,), generator=g)] = 1.0
    z = torch.randint(-1, 2, (32,), generator=g).float()
    first = tr.step(states, pi, z)
    for _ in range(12):
        last = tr.step(states, pi, z)
    assert last["loss"] < first["loss"] and tr.steps == 13
    # label-smoothed target: loss of the first step matches the formula evaluated by hand
    with pytest.raises(ValueError):
        tr.step(states, pi * 0.5, z)


def test_memmap_dataset_reads_the_sink_format(tmp_path):
    import pickle
    from chinesechesszero_amd.collect import TupleSink
    from chinesechesszero_amd.dataset import NpyMemmapDataset
    sink = TupleSink(str(tmp_path))
    s = (torch.rand(6, 17, 7, 10, 9) > 0.9).half()
    p = torch.rand(6, 2086)
    p = p / p.sum(1, keepdim=True)
    z = torch.tensor([1., -1, 0, 0, 1, -1])
    sink.append(s, p, z)
    sink.flush()
    ds = pickle.loads(pickle.dumps(NpyMemmapDataset(str(tmp_path))))
    assert len(ds) == 6
    st, pi, w = ds[4]
    assert st.dtype == torch.float16 and torch.equal(st, s[4]) and torch.allclose(pi, p[4]) and float(w) == 1.0
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False)
    b = next(iter(loader))
    assert b[0].shape == (4, 17, 7, 10, 9) and b[1].shape == (4, 2086) and b[2].shape == (4,)


def test_conv_entry_point_refuses_more_boards_than_its_32_bit_offsets_cover():
    """include/cczero.h: at most 93,206 boards per ccz_conv3x3_c256_f16 call (element offsets are 32-bit in the kernel).
    The check precedes any device work, so it is testable without a GPU: nothing is dereferenced."""
    from chinesechesszero_amd import _lib
    L = _lib.lib()
    fake = ctypes.c_void_p(0x10000)
    other = ctypes.c_void_p(0x20000)
    ok_boards, bad_boards = 93206, 93207
    assert ok_boards * 90 * 256 <= 2**31 - 1 < bad_boards * 90 * 256
    assert L.ccz_conv3x3_c256_f16(None, fake, fake, fake, None, other, bad_boards * 90, 1) == -1
    assert b"93206" in L.ccz_last_error()
    assert L.ccz_conv3x3_c256_f16(None, fake, fake, fake, None, other, 91, 1) == -1          # not a whole number of boards
    assert L.ccz_conv3x3_c256_f16(None, fake, fake, fake, None, fake, 90, 1) == -1           # output aliases the input
    assert L.ccz_conv3x3_c256_f16(None, ctypes.c_void_p(0x10008), fake, fake, None, other, 90, 1) == -1   # misaligned
    assert L.ccz_conv3x3_c256_f16(None, fake, fake, fake, None, other, 0, 1) == 0            # empty batch: nothing to do


def test_no_diagnostic_library_travels_with_the_product():
    """Only libcczero.so lives next to the package; diagnostic (-DCCZ_STAMPS) and A/B builds are made on demand by profiles/."""
    pkg = os.path.join(ROOT, "chinesechesszero_amd")
    assert sorted(f for f in os.listdir(pkg) if f.endswith(".so")) == ["libcczero.so"]


def test_viewer_window_serves_the_selected_board():
    """examples/viewer.py ChessWindow behind the hook (boardsvg.board_svg renders): update_board(svg, status) then GET /board returns the reference's JSON keys (frontend.py:120-136)."""
    import json
    import urllib.request
    from chinesechesszero_amd.boardsvg import board_svg
    from examples.viewer import ChessWindow
    sq = np.zeros(90, np.uint8)
    sq[4], sq[85], sq[0] = 7, 15, 3
    svg = board_svg(sq, last_move=(0, 9))
    assert svg.startswith("<svg") and svg.count("<circle") == 3 and ">K<" in svg and ">k<" in svg and ">R<" in svg
    w = ChessWindow("127.0.0.1", 0).start()
    try:
        w.update_board(svg, "to move: red - ply: 0")
        d = json.loads(urllib.request.urlopen(f"http://127.0.0.1:{w.port}/board", timeout=5).read())
        assert set(d) == {"svg", "status", "timestamp"} and d["svg"] == svg and d["status"].startswith("to move: red")
        page = urllib.request.urlopen(f"http://127.0.0.1:{w.port}/", timeout=5).read().decode()
        assert "/board" in page
        w.update_board("plain text board", "x")
        d = json.loads(urllib.request.urlopen(f"http://127.0.0.1:{w.port}/board", timeout=5).read())
        assert d["svg"].startswith("<pre>") and w.updates == 2
    finally:
        w.stop()


def test_parameters_are_the_references(golden):
    """parameters.py:1-28: every module-level constant the reference defines, with its value (dumped by make_golden.py)."""
    from chinesechesszero_amd import parameters
    ref = golden["meta"]["parameters"]
    assert ref["C_PUCT"] == 5 and ref["PLAYOUT"] == 1600
    assert {k: getattr(parameters, k) for k in ref} == ref


def test_the_package_asks_for_eight_hardware_queues_unless_the_user_chose():
    """chinesechesszero_amd/__init__.py: GPU_MAX_HW_QUEUES defaults to 8 (the tower's launch chains next to the exchange's streams:
    DESIGN section 7); a value the user exported wins."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import os, chinesechesszero_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    assert subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True).stdout.strip() == "8"
    env["GPU_MAX_HW_QUEUES"] = "2"
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True)
    assert r.stdout.strip() == "2" and "fewer than 8 hardware queues" in r.stderr      # respected, and said out loud
