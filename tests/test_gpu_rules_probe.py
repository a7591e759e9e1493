"""GPU: a cchess probe's golden file replayed on the PRODUCT -- the stateless movegen kernel behind the host ``Board``
(ccz_legal_moves) and the engine's own move-boundary kernel (k_finish_move: make-move, game end, winner) and selection-side
movegen (leaf id order).

    pytest -m gpu tests/test_gpu_rules_probe.py --rules-probe DIR

with DIR written by ``python tools/probe_cchess.py --out DIR`` where a real ``cchess`` is installed answers, in one command, "does
the engine match my cchess": ``legal_moves`` ORDER (net.py:154-157 -> mcts.py:37-39,47-48,59-61), PIECE_TYPES (tools.py:100), the
end / draw predicates (tools.py:119-123, mcts.py:116-126) and ``outcome().winner`` (game.py:208-219) on ~200 positions. Without the
option the probe runs here against the CPU oracle posing as ``cchess`` under the NON-canonical "python-chess-lineage" tables, so the
replay is exercised (tables installed from a preset file, type-major order, perpetual check) on every run -- that default proves the
mechanism, not parity with cchess (still unpinned: the module is not in this image).
"""
import importlib.util
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe_module():
    spec = importlib.util.spec_from_file_location("probe_cchess", os.path.join(ROOT, "tools", "probe_cchess.py"))
    P = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(P)
    return P


@pytest.fixture
def probe_files(rules_probe_dir, tmp_path):
    """(preset.json, cchess_golden.npz, came from a real cchess?)"""
    if rules_probe_dir is not None:
        return os.path.join(rules_probe_dir, "preset.json"), os.path.join(rules_probe_dir, "cchess_golden.npz"), True
    import oracle
    from chinesechesszero_amd import tools
    from fake_cchess import make_module
    lineage = tools.rule_presets()["python-chess-lineage"]
    oracle.set_rules(**lineage)
    try:
        _probe_module().probe(make_module("oracle"), str(tmp_path), n_games=6, plies=26)
    finally:
        oracle.set_rules()
    return str(tmp_path / "preset.json"), str(tmp_path / "cchess_golden.npz"), False


def _records(path):
    g = np.load(path)
    return g, json.loads(str(g["meta"]))


def _replay_on_host_board(preset, golden, real, allow_unsupported=False):
    """Every golden record against chinesechesszero_amd.game.Board (legal moves from ccz_legal_moves, ordered by the installed
    tables): ids IN ORDER, position, the four predicates, the winner. AssertionError names the first record that differs."""
    from fake_cchess import parse_fen
    from chinesechesszero_amd import tools
    from chinesechesszero_amd.game import Board
    tools.set_rules(preset=preset, allow_unsupported=allow_unsupported)     # raises if the probe found behaviours no table expresses
    try:
        g, meta = _records(golden)
        for j, rec in enumerate(meta):
            sq, red, half = parse_fen(rec["fen"])
            b = Board(sq, bool(red), half)
            for mv in rec["moves"]:
                b.push(mv)
            k = int(g["k"][j])
            assert b.legal_ids() == g["ids"][j][:k].tolist(), rec["label"]
            assert np.array_equal(b.squares(), g["squares"][j]) and int(b.turn) == int(g["turn"][j]), rec["label"]
            want = g["flags"][j].tolist()
            got = [int(b.is_game_over()), int(b.is_insufficient_material()), int(b.is_fourfold_repetition()), int(b.is_sixty_moves())]
            assert all(w < 0 or w == x for w, x in zip(want, got)), (rec["label"], want, got)     # -1 = the probed module could not say
            o = b.outcome()
            w = -2 if o is None else (-1 if o.winner is None else int(bool(o.winner)))
            assert w == int(g["winner"][j]), rec["label"]
        assert len(meta) > 140 or real
    finally:
        tools.set_rules()


def test_host_board_on_the_movegen_kernel_replays_the_probe(probe_files):
    _replay_on_host_board(*probe_files)


def _replay_on_engine(preset, golden, real, allow_unsupported=False):
    """The same records through the ENGINE: positions set with ccz_set_position, the recorded moves forced through k_finish_move one
    ply at a time (make-move, clock, repetition chain, game end, winner), then one selection on the fresh root for the kernel-side
    ``legal_moves`` order. over = is_game_over() or is_tie() (game.py:208); winner as game.py:210-219."""
    import oracle
    from fake_cchess import parse_fen
    from chinesechesszero_amd import _lib, tools
    from chinesechesszero_amd.engine import SelfPlayEngine
    tools.set_rules(preset=preset, allow_unsupported=allow_unsupported)
    L = oracle.lib()                    # (the checker's action table: uci -> id; no rules asked of it)
    names = {}
    for i in range(2086):
        f, t = L.xq_move_from(i), L.xq_move_to(i)
        names["abcdefghi"[f % 9] + str(f // 9) + "abcdefghi"[t % 9] + str(t // 9)] = i
    try:
        g, meta = _records(golden)
        by_len = {}
        for j, rec in enumerate(meta):
            by_len.setdefault(len(rec["moves"]), []).append(j)
        checked = ended = 0
        for n_moves, js in sorted(by_len.items()):
            e = SelfPlayEngine(len(js), n_playout=4, seed=1)     # takes the installed tables (tools.set_rules)
            for b, j in enumerate(js):
                sq, red, half = parse_fen(meta[j]["fen"])
                e.set_position(b, sq, 1 if red else 0, half)
            for t in range(n_moves):
                forced = np.array([names[meta[j]["moves"][t]] for j in js], np.int32)
                e.finish_move(forced_moves=forced, keep_tree=False)
            st = e.game_status()
            pos = e.root_positions()
            # after k_finish_move the game status says whether the game is over; a position that was SET (no move played yet) is
            # judged where the search meets it: the root is its own leaf, terminal by the same predicates (mcts.py:116-126)
            judged_by_search = n_moves == 0
            live = [b for b in range(len(js)) if not st["over"][b]]
            if live:
                e.select_leaves()      # a fresh root is its own leaf: the selection kernel's movegen lists its legal moves in order
                info = e.leaf_info()
            for b, j in enumerate(js):
                label = meta[j]["label"]
                fl = g["flags"][j].tolist()
                want_over = any(x > 0 for x in fl)
                gw = int(g["winner"][j])
                assert np.array_equal(pos[b][:90], g["squares"][j]), label
                if judged_by_search:
                    if want_over:      # LEAF_LOSS: the side to move has lost (mate and stalemate alike); LEAF_DRAW: a tie
                        lost = fl[0] > 0 and gw in (0, 1) and gw != int(g["turn"][j])
                        assert int(info["status"][b]) == (_lib.LEAF_LOSS if lost else _lib.LEAF_DRAW), (label, fl, gw)
                        ended += 1
                    else:
                        assert int(info["status"][b]) == _lib.LEAF_EXPAND, (label, fl)
                else:
                    assert bool(st["over"][b]) == want_over, (label, fl, meta[j]["moves"][-3:])
                    if want_over:
                        ended += 1
                        want_w = gw if (fl[0] > 0 and gw != -2) else -1          # a tie that is not game-over is a draw (game.py:208-219)
                        assert int(st["winner"][b]) == want_w, (label, fl, gw)
                if not want_over:
                    assert int(st["turn"][b]) == int(g["turn"][j]), label
                    k = int(g["k"][j])
                    assert int(info["k"][b]) == k and info["ids"][b][:k].tolist() == g["ids"][j][:k].tolist(), label
                checked += 1
            e.check_healthy()
            e.close()
        assert checked == len(meta) and (ended >= 8 or real)
    finally:
        tools.set_rules()


def test_engine_kernels_replay_the_probe(probe_files):
    _replay_on_engine(*probe_files)


def test_a_cchess_that_differs_makes_both_replays_fail(tmp_path):
    """The negative control: a "cchess" that scores stalemate as a draw and never claims the fourfold repetition (what an unmodified
    python-chess port would do). The probe says so (`unsupported_differences`), the product refuses the preset -- and when it is forced
    in, the golden file does NOT replay: the host Board and the engine kernels both name a record of exactly those endings. A replay
    that could not fail would not be evidence the day someone runs it against the real module."""
    from fake_cchess import make_module
    from chinesechesszero_amd import tools
    mod = make_module("oracle")
    Base = mod.Board

    class Lax(Base):
        def outcome(self):
            o = super().outcome()
            if o is not None and not self.legal_moves and not self.b.in_check():
                o.winner = None          # stalemate = draw
            return o

        def is_fourfold_repetition(self):
            return False                 # never claims the draw at four occurrences

    mod.Board = Lax
    got, golden = _probe_module().probe(mod, str(tmp_path), n_games=2, plies=8)
    assert got["unsupported_differences"]
    files = (str(tmp_path / "preset.json"), str(tmp_path / "cchess_golden.npz"), True)
    with pytest.raises(ValueError, match="no table expresses"):
        _replay_on_host_board(*files)
    for replay in (_replay_on_host_board, _replay_on_engine):
        with pytest.raises(AssertionError) as err:
            replay(*files, allow_unsupported=True)
        assert any(word in str(err.value) for word in ("stalemate", "repetition", "fourfold", "perpetual")), str(err.value)[:300]
    assert tools.PRESET == "canonical"                                          # (the helpers restore the defaults)


def test_product_refuses_a_preset_with_unsupported_differences(tmp_path):
    from chinesechesszero_amd import tools
    p = {"schema": 1, "plane_of_type": [0, 0, 1, 2, 3, 4, 5, 6], "type_rank": None, "move_rank": None, "pawn_move_resets_clock": False,
         "perpetual_check": False, "unsupported_differences": ["is_fourfold_repetition() never true"]}
    path = tmp_path / "preset.json"
    path.write_text(json.dumps(p))
    try:
        with pytest.raises(ValueError, match="no table expresses"):
            tools.set_rules(preset=str(path))
        assert tools.PRESET == "canonical"
        tools.set_rules(preset=str(path), allow_unsupported=True)
        assert tools.PRESET == str(path)
    finally:
        tools.set_rules()
