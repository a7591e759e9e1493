"""GPU: bench.py as the driver runs it -- the one-line JSON contract at N = 1, and the N = 2 control flow rehearsed
with two ranks sharing the one GPU of this box over gloo (RCCL refuses two ranks on one device; the real N > 1 run is
``--backend nccl``, one rank per GPU, launched by the driver).

This file sorts first on purpose: its tests START OTHER PROGRAMS, which a process that has already initialised the
GPU must not do on this pool. Nothing here touches HIP in the pytest process itself.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--boards", "256", "--blocks", "2", "--channels", "256", "--preroll-plies", "12", "--max-plies", "12"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _json_line(out: str) -> dict:
    lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, out[-3000:]
    return json.loads(lines[0])


def _env():
    env = dict(os.environ)
    env["CCZ_MIOPEN_FIND"] = "0"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


@pytest.mark.parametrize("exchange", ["async", "sync"])
def test_bench_starts_its_own_ranks_and_exchanges_inside_the_timed_window(exchange):
    """`python bench.py --gpus 2 ...` with NO launcher (the driver's N = 1 command shape with another N): bench.py starts its
    own two ranks as a child `torch.distributed.run` before touching the GPU and relays rank 0's line and the exit code.
    async (the default): replay.AsyncRecordExchange -- post at the boundary, the collective issued by a later step, the drain inside
    the timed region; sync: round 4's blocking all-gather at the boundary."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "2",
           "--backend", "gloo", "--share-gpu", "--gather-plies", "1024", "--playout", "16", "--exchange", exchange] + SMALL
    env = _env()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["steps"] == 12 and j["warmup"] == 2 and j["scaling"] == "weak" and j["unit"] == "sims/s"
    m = j["multi_gpu"]
    assert m["world_size"] == 2 and m["ranks_seen"] == 2 and m["backend"] == "gloo" and m["exchange"] == exchange
    assert j["deviations"]["scope"] == "summed over all ranks" and j["deviations"]["pruned_subtrees_total"] == 0
    assert m["boards_per_rank"] == [256, 256] and m["board_id_base_per_rank"] == [0, 256]
    assert len(m["rank_step_ms"]) == 2 and all(0 < v <= j["ms_per_step"] * 1.001 for v in m["rank_step_ms"])
    # a 12-step window of a 16-simulation move still holds one real move boundary with its exchange
    assert m["exchanges_in_window"] == 1
    if exchange == "sync":
        assert m["collectives_in_window"] == 1 and m["gather_ms"] > 0      # ONE blocking collective at the boundary
    else:
        # the boundary's records travel in ONE collective issued by a later step -- or, when the K steps end first, by the drain (whose
        # first exchange then carries them with the closing flag up); the drain never needs more than two
        assert m["collectives_in_window"] + m["collectives_in_drain"] == 1
        assert m["exchanges_completed_in_window_and_drain"] == m["collectives_in_window"] + m["collectives_in_drain"]
        assert m["exchange_host_ms_rank0"] >= 0 and len(m["exchange_max_call_ms_per_rank"]) == 2
        assert m["plies_sent_rank0"] == j["move_boundary"]["rows_harvested_rank0"] // 2
    # boards 0, 12, 24, ... of each rank stand at the 12-ply cap: 22 games x 12 plies x 2 (mirror images) per rank at least
    assert m["rows_gathered"] >= 2 * 22 * 12 * 2 and j["move_boundary"]["games_finished"] >= 44
    # the wire carries compact ply records: 880 B per ply (= two dense rows of 29,768 B), one slot per rank: sync = the fixed full slot;
    # async = the smallest power-of-two class (>= 64 plies) that holds the largest announcement of that exchange
    assert m["gather_capacity_plies"] == 1024
    if exchange == "sync":
        assert m["bytes_sent_per_rank_per_collective"] == 64 + 1024 * 880
    else:
        cls = (m["bytes_sent_per_rank_per_collective"] - 64) // 880
        assert cls in (64, 128, 256, 512, 1024) and m["bytes_sent_per_rank_per_collective"] == -(-(64 + cls * 880) // 64) * 64
        assert cls >= m["plies_sent_rank0"] > cls // 4        # (both ranks finish about the same number of games)
        # the drain's closing exchange was decided from the announcements alone: no collective
        assert m["exchanges_without_a_collective"] >= 1 and m["exchanges_decided"] == m["exchanges_without_a_collective"] + m["collectives_in_window"] + m["collectives_in_drain"]
    assert m["payload_bytes_rank0_per_exchange"] == 880 * j["move_boundary"]["rows_harvested_rank0"] // 2
    # every rank's ring received the union: the window's exchange and those of the two untimed warm-up moves before it
    assert m["bad_records"] == 0 and m["replay_rows_total"] >= m["rows_gathered"] and j["config"]["warm_moves"] == 2
    assert m["error_flags_any"] == 0 and m["ring_ranks"] == "all" and m["expand_ms_host"] >= 0 and m["dist_timeout_s"] == 180
    assert len(m["per_rank_sims_per_sec"]) == 2 and all(v > 0 for v in m["per_rank_sims_per_sec"])
    # whole-job value = all ranks' simulations over the slowest rank's time: never above the sum of the per-rank rates
    assert 0 < j["value"] <= sum(m["per_rank_sims_per_sec"]) * 1.001
    assert j["move_boundary"]["in_window"] == 1 and j["move_boundary"]["ms_host"] > 0
    assert abs(j["ms_per_step"] * 12 * 1e-3 * j["value"] - 2 * 256 * 12) < 1e-3 * 2 * 256 * 12


def _two_ranks(extra, steps="12"):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", steps, "--warmup", "2",
           "--backend", "gloo", "--share-gpu", "--gather-plies", "1024", "--playout", "16"] + extra + SMALL
    env = _env()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    return _json_line(r.stdout)


def test_bench_a_slow_rank_delays_data_not_its_peer():
    """Rank 1 sleeps 3 s at the window's move boundary (--slow-rank 1:3). With the asynchronous exchange rank 0's own K steps
    take what they take without it (the reference's collectors never wait for each other, collect.py:181-183); with round 4's
    blocking all-gather rank 0 sat in the collective until rank 1 woke up. The job's time (slowest rank) holds the sleep either way."""
    a = _two_ranks(["--slow-rank", "1:3"])
    m = a["multi_gpu"]
    own = [v * 12 * 1e-3 for v in m["rank_step_ms"]]       # each rank's own time over its 12 steps, seconds
    assert own[1] >= 3.0 and own[0] < own[1] - 2.5, own   # rank 0 did not wait for the sleeper
    assert a["ms_per_step"] * 12 * 1e-3 >= 3.0            # the job did (max over ranks + drain)
    assert m["slow_rank"] == {"rank": 1, "sleep_s_per_boundary": 3.0}
    assert m["rows_gathered"] >= 2 * 22 * 12 * 2 and m["bad_records"] == 0     # and nothing was lost: the drain delivered rank 1's games
    s = _two_ranks(["--slow-rank", "1:3", "--exchange", "sync"])
    own = [v * 12 * 1e-3 for v in s["multi_gpu"]["rank_step_ms"]]
    assert own[0] >= 2.5, own                             # the control: the blocking exchange couples the ranks


def test_bench_four_ranks_rehearsal_one_gpu():
    """Four ranks over gloo on the one GPU. (Eight cannot be rehearsed on a GPU box: its process guard allows six processes on the
    card, and a first attempt with six ranks was killed at seven -- the ranks plus one launcher-side process. The world-8 control flow
    of the exchange -- uneven loads, an empty rank, an aborting rank -- is tests/test_cpu_async_exchange.py.) Every rank seen, rows of
    all four ranks in every ring, one collective per exchange, no error flag."""
    R = 4
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(R), "--steps", "40", "--warmup", "2", "--backend", "gloo", "--share-gpu",
           "--gather-plies", "1024", "--playout", "16", "--boards", "128", "--blocks", "2", "--channels", "256", "--preroll-plies", "12", "--max-plies", "12"]
    env = _env()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    j = _json_line(r.stdout)
    m = j["multi_gpu"]
    assert j["n_gpus"] == R and m["ranks_seen"] == R and m["error_flags_any"] == 0 and m["bad_records"] == 0
    assert m["boards_per_rank"] == [128] * R and m["board_id_base_per_rank"] == [128 * r for r in range(R)]
    # 40 steps of 16-simulation moves: 2-3 move boundaries; every one of them posts, every exchange is ONE collective
    assert m["exchanges_in_window"] in (2, 3)
    assert m["exchanges_completed_in_window_and_drain"] == m["collectives_in_window"] + m["collectives_in_drain"] >= 1
    # per boundary every rank finishes >= 10 games of 12 plies (boards 0, 12, ... at the cap): the ring holds all ranks' rows
    assert m["rows_gathered"] >= m["exchanges_in_window"] * R * 10 * 12 * 2 and m["replay_rows_total"] >= m["rows_gathered"]
    assert m["games_gathered"] >= m["exchanges_in_window"] * R * 10
    assert len(m["rank_step_ms"]) == R and max(m["exchange_max_call_ms_per_rank"]) < 2000


def test_bench_a_rank_that_fails_inside_the_timed_window_ends_the_job_fast():
    """Rank 1 raises at its 3rd timed step (--inject-fault) while rank 0 goes on into the move boundary's all-gather. The rank
    guard ends rank 1's process at once, the launcher tears the job down: non-zero exit code, the failing rank named, no JSON
    line -- long before the 600-second collective timeout this run asks for."""
    import time
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "2", "--backend", "gloo",
           "--share-gpu", "--gather-plies", "1024", "--playout", "16", "--inject-fault", "1:3", "--dist-timeout", "600"] + SMALL
    env = _env()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    took = time.time() - t0
    assert r.returncode != 0 and "[rank 1] failed" in r.stderr and "injected fault on rank 1" in r.stderr, (r.stdout[-1000:], r.stderr[-3000:])
    assert '"metric"' not in r.stdout
    assert took < 240, took      # set-up of two ranks + a few steps; nowhere near the 600 s a hung collective would take


def test_bench_refuses_more_ranks_than_gpus_before_starting_any():
    """`python bench.py --gpus 8` on a one-GPU box: the parent counts the GPUs without touching HIP and says no in one line."""
    import time
    env = _env()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"] + SMALL, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "not starting any rank" in r.stderr and "--gpus 8 needs 8 visible GPU(s)" in r.stderr
    assert time.time() - t0 < 120 and '"metric"' not in r.stdout


def test_bench_under_the_launcher_with_one_rank_prints_the_single_gpu_line():
    """SCALE's N = 1 (`torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`) has to be BENCH's N = 1 line: the same keys
    (roofline, cpu_baseline, no multi_gpu block), the same workload description, n_gpus 1."""
    args = ["--steps", "10", "--warmup", "3", "--cpu-baseline-seconds", "2", "--playout", "64"] + SMALL
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert plain.returncode == 0, (plain.stdout[-2000:], plain.stderr[-3000:])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args
    under = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert under.returncode == 0, (under.stdout[-2000:], under.stderr[-3000:])
    a, b = _json_line(plain.stdout), _json_line(under.stdout)
    assert set(a) == set(b) and "multi_gpu" not in b and "cpu_baseline" in b and b["n_gpus"] == 1
    assert a["config"] == b["config"] and set(a["roofline"]) == set(b["roofline"]) and set(a["cpu_baseline"]) == set(b["cpu_baseline"])
    # the same deterministic search: identical tree statistics in both runs
    assert a["nodes_peak"] == b["nodes_peak"] and a["roofline"]["k_bar"] == b["roofline"]["k_bar"] and a["roofline"]["d_bar"] == b["roofline"]["d_bar"]


def test_bench_group_of_one_on_rccl_and_the_graph_replayed_simulator_line():
    """Two command shapes the round-5 evidence rests on, kept alive at test size: `--rccl-group-of-one` (the whole N > 1 path -- process
    group on backend nccl, asynchronous exchange, replay ring -- with ONE rank, i.e. real RCCL collectives on the one GPU) and
    `--evaluator stub --graph` (the simulator alone, evaluator + k_step replayed as one captured hipGraph per simulation)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "24", "--warmup", "2", "--playout", "16", "--rccl-group-of-one",
                        "--gather-plies", "1024", "--no-cpu-baseline"] + SMALL, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    j = _json_line(r.stdout)
    m = j["multi_gpu"]
    assert j["n_gpus"] == 1 and m["world_size"] == 1 and m["backend"] == "nccl" and "rccl_group_of_one" in m and j["gpu_max_hw_queues"] == "8"
    assert m["collectives_in_window"] + m["collectives_in_drain"] >= 1 and m["rows_gathered"] >= 22 * 12 * 2 and m["bad_records"] == 0
    assert m["replay_rows_total"] >= m["rows_gathered"] and m["error_flags_any"] == 0 and "cpu_baseline" not in j
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "48", "--warmup", "2", "--playout", "16", "--evaluator", "stub", "--graph",
                        "--no-cpu-baseline"] + SMALL, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    j = _json_line(r.stdout)
    rf = j["roofline"]
    assert j["config"]["evaluator"] == "stub" and "hipGraph replay" in rf["duration_source"] and rf["avg_launch_us_hip_events_raw"] is None
    assert 0 < rf["frac"] < 1 and j["move_boundary"]["in_window"] == 3 and j["error_flags_any"] == 0
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--graph"] + SMALL, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--graph is for the simulator-only line" in (r.stdout + r.stderr)


def test_bench_refuses_a_gpu_count_that_contradicts_the_launcher():
    env = _env()
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stdout + r.stderr)


def test_bench_two_ranks_with_concurrent_trainer_on_rank0():
    """BASELINE configs[4]'s control flow (N ranks of self-play, all-gather of the tuples, a trainer on rank 0 consuming the
    gathered rows from its replay buffer on a side stream) rehearsed with two ranks on the one GPU."""
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "16", "--warmup", "2",
           "--backend", "gloo", "--share-gpu", "--gather-plies", "1024", "--playout", "32", "--train-every", "4", "--boards-rank0", "auto"] + SMALL
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    j = _json_line(r.stdout)
    assert j["multi_gpu"]["ranks_seen"] == 2 and j["multi_gpu"]["exchanges_in_window"] == 1
    # rank 0 shares its GPU with the trainer: it calibrated itself a lighter load, the global board ids stay a prefix sum
    bpr = j["multi_gpu"]["boards_per_rank"]
    cal = j["rank0_calibration"]
    assert bpr[0] < bpr[1] == 256 and bpr[0] == cal["boards_rank0"] and bpr[0] % 64 == 0 and j["multi_gpu"]["board_id_base_per_rank"] == [0, bpr[0]]
    # measured on the real workload: a full engine's step without and with the trainer, then the step of the engine that runs
    assert 0 < cal["plain_step_ms"] < cal["step_ms_with_trainer_at_full_boards"] and cal["trainer_ms_per_step"] > 0
    assert cal["step_ms_with_trainer_at_boards_rank0"] < cal["step_ms_with_trainer_at_full_boards"]
    assert abs(j["ms_per_step"] * 16 * 1e-3 * j["value"] - sum(bpr) * 16) < 1e-3 * sum(bpr) * 16
    # both ranks' finished games reached rank 0's ring: boards 0, 12, 24, ... of each rank stand at the 12-ply cap
    assert j["multi_gpu"]["rows_gathered"] >= sum((b + 11) // 12 for b in bpr) * 12 * 2
    assert j["trainer_updates"] == 4                               # one 2048-row update per 4 steps, inside the window
    assert j["multi_gpu"]["ring_ranks"] == "0"                     # with a trainer only its rank expands the records into a dense ring
    assert j["value"] > 0


def test_bench_single_gpu_line_contract():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--cpu-baseline-seconds", "3", "--playout", "64"] + SMALL
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    j = _json_line(r.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 10 and j["warmup"] == 3 and j["higher_is_better"] is True and j["vs_baseline"] is None
    assert "workload" in j["config"] and "model" not in j["config"]
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["traffic"] is None or "replayed" in rf["traffic_source"]      # committed counters are labelled as such
    assert rf["d_bar"] > 1.0                                                # mid-move trees, not the first simulations of move 1
    mb = j["move_boundary"]
    assert mb["in_window"] == 1 and mb["games_finished"] >= 22 and mb["rows_harvested_rank0"] >= 22 * 12 * 2
    assert mb["ms_events"] > 0 and mb["ms_host"] > 0 and j["moves_per_sec"] > 0
    # this small workload is not the profiled one: the fraction is THIS run's raw HIP-event figure (nothing subtracted: a lower bound),
    # and the line says which code it ran and why the committed profile does not apply
    assert "LIVE" in rf["duration_source"] and "RAW HIP events" in rf["duration_source"] and "another workload" in rf["committed_profile_note"]
    assert rf["avg_launch_us"] == rf["avg_launch_us_hip_events_raw"] > 0 and rf["frac"] == rf["frac_hip_events_raw"]
    assert rf["frac_at_committed_rocprofv3_duration"] is None and rf["avg_launch_us_committed_rocprofv3"] is None
    assert len(rf["code_hash"]) == 16 and "profile_head" in rf and rf["live_over_profile"] is None
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    # the split net / rules / tree of a CPU playout, and the same loop with a constant-time stub evaluator
    sp = cb["split"]
    assert 0.5 < sp["net"] < 1.0 and 0 < sp["rules"] < 0.2 and 0 < sp["tree"] < 0.2 and abs(sp["net"] + sp["rules"] + sp["tree"] + sp["python_glue"] - 1.0) < 1e-6
    assert cb["stub"]["value"] > 20 * cb["value"] and cb["stub"]["unit"] == "sims/s"
    nr = j["net_roofline"]
    assert nr["bound"] == "mfma" and nr["peak"] == 2500.0 and 0 < nr["frac"] < 1
    # the tower figure describes the timed window: its launches fit in the step they are part of; the post-window full-batch
    # figure is kept under its own name
    assert "in the window" in nr["duration_source"] and nr["launches_per_step"] * nr["avg_launch_us"] <= j["ms_per_step"] * 1e3
    assert nr["rows_per_launch"] <= 256 and 0 < nr["frac_full_batch"] < 1
    assert j["error_flags_any"] == 0
    dv = j["deviations"]                     # where the engine departs from the reference, counted in the line itself
    assert dv["pruned_subtrees_in_window"] == 0 and dv["pruned_subtrees_total"] == 0 and dv["cache_verify_mismatches"] is None
    assert 0 <= dv["truncated_games_in_window"] <= dv["truncated_games_total"] <= dv["games_finished_total_rank0"] and dv["max_plies"] > 0
    assert dv["scope"] == "this GPU"
    assert j["plies"]["start_mean"] > 3


_RCCL_PROBE = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1])
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from chinesechesszero_amd.replay import RecordGatherer, ReplayBuffer, TupleGatherer, exchange_finished_games
from chinesechesszero_amd.selfplay import BatchedSelfPlay
from chinesechesszero_amd.engine import expand_records
from chinesechesszero_amd.net import uniform_evaluator
def play(seed):
    sp = BatchedSelfPlay(uniform_evaluator, 16, n_playout=3, seed=seed, max_plies=4)
    for _ in range(5):
        sp.run_move()
    assert sp.engine.game_status()["over"].all()
    return sp
# dense rows through the fused buffer (round 2's wire format)
g = TupleGatherer(96, dev, always_collective=True)
sp = play(1)
rows = 0
dense = []
it = iter(sp.harvest_chunks(g.cap))
chunk = next(it)
while chunk is not None:
    nxt = next(it, None)
    s, p, z = g.gather(*chunk, more=nxt is not None, user=16 if rows == 0 else 0)
    assert g.collectives == 1 and g.rows_per_rank == [int(chunk[2].shape[0])] and g.any_more == (nxt is not None)
    assert torch.equal(s, chunk[0]) and torch.equal(p, chunk[1]) and torch.equal(z, chunk[2])   # bytes survive the fused buffer
    dense.append((s.clone(), p.clone(), z.clone()))
    rows += int(z.shape[0])
    chunk = nxt
assert rows == 16 * 4 * 2, rows
big = torch.rand((250, 2086), device=dev)
S, P, Z = g.gather(torch.zeros((250, 17, 7, 10, 9), dtype=torch.float16, device=dev), big, torch.arange(250, device=dev).float())
assert g.collectives == 3 and torch.equal(P, big) and torch.equal(Z, torch.arange(250, device=dev).float())   # 250 rows through 96-row rounds
# the same games as COMPACT RECORDS through RCCL, 24 plies per round (whole games only), expanded into a replay ring:
# byte for byte the dense rows above
rg = RecordGatherer(24, dev, always_collective=True)
sp2 = play(1)
rb = ReplayBuffer(200, dev)
n = colls = 0
for union, games in exchange_finished_games(sp2, rg, 16):
    assert union.shape[0] <= 24 and union.shape[0] % 4 == 0 and rg.collectives == 1
    n += rb.append_records(union, sp2.engine.record_flags(), sp2.engine.plane_of_type)
    colls += 1
assert n == 128 and colls == 3 and rb.total == 128     # 64 plies in rounds of 24, 24, 16
D = [torch.cat([d[i] for d in dense]) for i in range(3)]
assert torch.equal(rb.states[:128], D[0]) and torch.equal(rb.pi[:128], D[1]) and torch.equal(rb.z[:128], D[2])
# the ASYNCHRONOUS exchange on RCCL (round 5): async_op all_gather_into_tensor issued from the side stream, completion polled with
# is_completed(), headers read on the side stream, the store handshake -- while the main stream keeps working; same bytes again
from chinesechesszero_amd.replay import AsyncRecordExchange
import time
ax = AsyncRecordExchange(24, dev, always_collective=True, timeout_s=60)
sp3 = play(1)
rb2 = ReplayBuffer(200, dev)
busy = torch.zeros((4096, 4096), device=dev)
got = []
def take(done):
    for x in done:
        got.append((x.index, int(x.union.shape[0]), x.games))
        rb2.append_records(x.union, sp3.engine.record_flags(), sp3.engine.plane_of_type)
take(ax.post(list(sp3.harvest_record_chunks(24)), games=16))      # 64 plies against a 24-ply slot: the backlog carries over
t0 = time.perf_counter()
while ax.completed < 3 and time.perf_counter() - t0 < 30:
    busy.add_(1.0)                                                 # main-stream work between the ticks
    take(ax.tick())
    if ax._work is None and ax._backlog_plies:
        take(ax.post([], games=0))                                 # the next boundary: announce the next exchange
for x in ax.flush_iter():
    take([x])
torch.cuda.synchronize()
assert sum(g[1] for g in got) == 64 and sum(g[2] for g in got) == 16 and [g[0] for g in got] == [0, 1, 2], got
assert all(g[1] <= 24 and g[1] % 4 == 0 for g in got)
assert ax.collectives == 3 and ax.virtual >= 1 and ax.issued == ax.completed == ax.collectives + ax.virtual    # the closing exchange: no collective
assert rb2.total == 128 and torch.equal(rb2.states[:128], D[0]) and torch.equal(rb2.pi[:128], D[1]) and torch.equal(rb2.z[:128], D[2])
assert ax.max_call_s < 5.0
t = torch.ones(1, dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
dist.destroy_process_group()
print("RCCL_PROBE_OK", rows)
"""


def test_fused_gather_on_the_rccl_backend_group_of_one():
    """The collective itself on backend nccl (= RCCL) with real device buffers: uint8 all_gather_into_tensor of the fused
    buffer, the pinned header copy, float64 all-reduce and barrier as bench.py issues them. A group of one is all a single
    GPU allows (RCCL refuses two ranks on one device); the N > 1 control flow is the gloo rehearsal above."""
    r = subprocess.run([sys.executable, "-c", _RCCL_PROBE, str(_free_port())], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_PROBE_OK 128" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_the_exchange_streams_do_not_serialise_the_towers_launch_chains():
    """Found in round 5 by running the N>1 path on RCCL in a group of one at full size: the tower's three launch chains shared hardware
    queues with the process group's stream and the exchange's side stream and ran one after the other -- 26.6 ms per step instead of
    21.2 (profiles/r05_g1_*.json, r05_h_*.json, r05_hwq_probe.txt). Two remedies, both checked here through profiles/hwq_probe.py (the
    40x256 evaluator on 4096 rows in a process of its own): the package asks for eight hardware queues at import, and the launch-chain
    streams are bound when the first inference copy is built (net.chain_streams), before the exchange is first used."""
    def run(*cfg, queues=None):
        env = _env()
        env.pop("GPU_MAX_HW_QUEUES", None)
        if queues:
            env["GPU_MAX_HW_QUEUES"] = queues
        r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "hwq_probe.py")] + list(cfg), cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("HWQ_PROBE")][0][len("HWQ_PROBE "):])
        d["ms"] = min(d["ms"])
        return d
    alone = run("0", "none")
    bench_order = run("1", "eval_first")               # bench.py: process group, first evaluation, then the first move boundary's exchange
    cli_order = run("1", "net_then_exchange")          # collector CLI: process group, net built, weights broadcast / exchange, first evaluation
    assert alone["queues"] == bench_order["queues"] == cli_order["queues"] == "8"      # the package's default reached the HIP runtime's environment
    assert bench_order["ms"] < 1.03 * alone["ms"] and cli_order["ms"] < 1.03 * alone["ms"], (alone, bench_order, cli_order)
    four = run("1", "eval_first", queues="4")          # the control: HIP's default (reported, not asserted: driver versions may differ)
    print("evaluator ms: alone %.2f; behind a process group + used exchange, bench order %.2f, CLI order %.2f; bench order with four queues %.2f"
          % (alone["ms"], bench_order["ms"], cli_order["ms"], four["ms"]))


def test_collect_cli_writes_the_trainer_files(tmp_path):
    """`python -m chinesechesszero_amd.collect` as a user runs it (reference collect.py:188-198 CLI: --show / --model; here also
    --boards / --playout / --moves ...): lockstep self-play on the GPU, harvest, shards, and on exit the converter step."""
    cmd = [sys.executable, "-m", "chinesechesszero_amd.collect", "--boards", "32", "--playout", "4", "--blocks", "1", "--channels", "32",
           "--max-plies", "5", "--moves", "13", "--model", "no_such_model.pkl", "--data-dir", str(tmp_path / "data")]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    import numpy as np
    meta = json.load(open(tmp_path / "data" / "meta.json"))
    # 13 moves with a 5-ply cap: every board is adjudicated at its 6th and 12th move -> 2 games x 5 plies x 2 (mirror) per board
    assert meta["total_count"] == 32 * 2 * 5 * 2 and meta["iters"] == 64 and meta["mcts_dtype"] == "float64"
    states = np.load(tmp_path / "data" / "states.npy", mmap_mode="r")
    pi = np.load(tmp_path / "data" / "mcts.npy", mmap_mode="r")
    z = np.load(tmp_path / "data" / "winners.npy", mmap_mode="r")
    assert states.shape == (640, 17, 7, 10, 9) and states.dtype == np.float16 and pi.shape == (640, 2086) and z.shape == (640,)
    assert np.allclose(np.asarray(pi).sum(1), 1.0, atol=1e-5) and float(np.abs(np.asarray(z)).max()) == 0.0
    assert not [f for f in os.listdir(tmp_path / "data") if f.startswith(".shard_")]


def test_collect_cli_two_ranks_one_store(tmp_path):
    """The reference's N collectors (README.md:31-48: N shell commands appending to one file) as ONE job: `torch.distributed.run
    --nproc-per-node 2 -m chinesechesszero_amd.collect`. Both ranks play their boards with rank 0's weights, finished games travel
    through the asynchronous exchange, rank 0 stores the union and writes the trainer's files; rank 1 stores nothing."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "chinesechesszero_amd.collect", "--boards", "32", "--playout", "4", "--blocks", "1",
           "--channels", "32", "--max-plies", "5", "--moves", "13", "--model", "no_such_model.pkl", "--data-dir", str(tmp_path / "data"),
           "--backend", "gloo", "--share-gpu"]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    import numpy as np
    meta = json.load(open(tmp_path / "data" / "meta.json"))
    # 13 moves with a 5-ply cap: every board of BOTH ranks is adjudicated twice -> 2 ranks x 32 boards x 2 games x 5 plies x 2 (mirror)
    assert meta["total_count"] == 2 * 32 * 2 * 5 * 2 and meta["iters"] == 128
    pi = np.load(tmp_path / "data" / "mcts.npy", mmap_mode="r")
    assert pi.shape == (1280, 2086) and np.allclose(np.asarray(pi).sum(1), 1.0, atol=1e-5)
    assert not os.path.exists(tmp_path / "data" / ".rank1" / "states.npy")       # the union lives in ONE store


def test_uci_cli_over_stdin_with_the_real_net():
    """`python -m chinesechesszero_amd.uci` (the README's "standard UCI protocol", README.md:3, which the reference never
    implemented): a session over stdin/stdout with the default 40x256 net, single-board search replayed as a hipGraph."""
    from oracle import OracleBoard
    script = "\n".join(["uci", "isready", "ucinewgame", "position startpos moves h2e2 h9g7", "go nodes 24", "d", "quit"]) + "\n"
    r = subprocess.run([sys.executable, "-m", "chinesechesszero_amd.uci"], input=script, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    out = r.stdout.splitlines()
    assert "uciok" in out and "readyok" in out
    best = [l for l in out if l.startswith("bestmove")]
    assert len(best) == 1 and any(l.startswith("info nodes 24 ") for l in out)
    b = OracleBoard()          # the CPU checker: nothing in this file may touch the GPU in the pytest process itself
    b.push("h2e2")
    b.push("h9g7")
    assert best[0].split()[1] in b.legal_moves
