#!/usr/bin/env python3
"""bench.py -- self-play MCTS throughput of the MI355X lockstep engine (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]      (N > 1 without a launcher: bench.py starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one lockstep simulation of every board: select+make-move+movegen+encode (HIP) ->
policy-value net (PyTorch-ROCm fp16, same stream) -> expand+backup (HIP); every ``n_playout``-th
step also plays one move on every board (pi, Dirichlet-mixed choice, re-root, game end; HIP), harvests
the training rows of the games that ended and exchanges them (RCCL all-gather when N > 1). Workload =
BASELINE.json configs[2]: 4096 concurrent boards per GPU x 400 sims/move, Dirichlet root noise on,
random-init 40x256 net (synthetic).

What the K timed steps see (untimed setup before them): the boards are in the steady state of continuous
self-play -- spread evenly over plies 1..200 of their games, diverged positions -- and the trees have
been searched with the real evaluator up to simulation n - K/2 of the current move, so the timed window
holds the end of one move's search, a REAL move boundary (finish_move + harvest + restart [+ all-gather])
and the start of the next move on the kept subtrees, whatever K is. Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import chinesechesszero_amd  # noqa: E402,F401  (first: sets GPU_MAX_HW_QUEUES before anything initialises HIP -- see its __init__)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md (6.29e12 measured copy peak)
MFMA_PEAK_F16 = 2.5e15  # FLOP/s dense fp16/bf16, MI355X_MICROARCH.md (never the 2:1-sparsity figure)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--boards", type=int, default=4096, help="concurrent boards per GPU")
    ap.add_argument("--playout", type=int, default=400, help="simulations per move")
    ap.add_argument("--evaluator", choices=["net", "stub"], default="net")
    ap.add_argument("--blocks", type=int, default=40)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-every", type=int, default=0, help="BASELINE config 5: rank 0 runs one trainer update "
                    "(batch 2048, side stream, GPU0) every this many simulation steps, fed from the replay buffer")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse "
                    "the N>1 control flow with several ranks sharing one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--preroll-plies", type=int, default=200, help="untimed setup: spread the boards over plies 1..P of their games "
                    "(stub evaluator, uniformly random moves; 0 = all boards on the opening position, as round 1 timed it)")
    ap.add_argument("--max-plies", type=int, default=200, help="games are adjudicated as draws at this many plies (the soak's cap); "
                    "with --preroll-plies equal to it every move ends ~boards/P games, so every move boundary harvests real rows")
    ap.add_argument("--no-align", dest="align", action="store_false", help="do not advance the search to the point where the timed "
                    "window straddles a move boundary (the window then starts at the first simulation of a move)")
    ap.add_argument("--align-evaluator", choices=["net", "stub"], default="net", help="evaluator of the untimed alignment steps: the real "
                    "net (default) or the stub (fast; for rocprofv3 PMC passes, where every kernel of an untimed step costs profiler time)")
    ap.add_argument("--gather-plies", type=int, default=32768, help="N>1: ply capacity of the fused all-gather slot (880 B per ply record; "
                    "one move of 4096 boards finishes ~14 k plies = ~28 k dense rows; what does not fit waits for the next exchange)")
    ap.add_argument("--exchange", choices=["async", "sync"], default="async", help="N>1: async (default) = replay.AsyncRecordExchange: no rank "
                    "ever waits for another one (records join a backlog at the move boundary, the all-gather is issued from a side stream once "
                    "every rank has announced it, its result is picked up by a later step); sync = round 3/4's blocking all-gather at every move boundary")
    ap.add_argument("--boards-rank0", default="", help="N>1 with --train-every: boards of rank 0, the rank that shares its GPU with the trainer "
                    "(a number, or 'auto' = calibrated before the window so that rank 0's step with the trainer takes as long as a plain "
                    "rank's step); the other ranks keep --boards. Global board ids are a prefix sum: every board keeps its RNG stream")
    ap.add_argument("--graph", action="store_true", help="simulator-only runs (--evaluator stub, one GPU): replay evaluator + k_step as ONE captured hipGraph "
                    "per simulation (BatchedSelfPlay(use_graph=True)) without per-step HIP events -- the Python launch loop with its events costs "
                    "~75 us per step, k_step ~30: this is the device-bound figure")
    ap.add_argument("--rccl-group-of-one", action="store_true", help="one GPU: run the N>1 exchange path for real on backend nccl (= RCCL) in a process "
                    "group of ONE rank -- the asynchronous all-gather from the side stream, the store handshake, the expansion into the replay ring, "
                    "inside the full-size timed loop (what one GPU can show of the exchange's cost on the rank that issues it)")
    ap.add_argument("--slow-rank", default="", help="testing: RANK:SECONDS -- that rank sleeps this long at every move boundary of the timed window "
                    "(with --exchange async its peers' step rates must not change)")
    ap.add_argument("--replay-rows", type=int, default=0, help="N>1 / trainer: rows of the dense replay ring every rank keeps in HBM "
                    "(0 = 40,000 x world size: more than one move's rows of all ranks)")
    ap.add_argument("--eval-cache-log2", type=int, default=24, help="evaluation cache of 2^n positions (528 B each; 0 = none): leaves whose "
                    "position was evaluated before -- by this board, another board, or another board of the same step -- skip the "
                    "network (the reference evaluates every leaf, mcts.py:114; results are identical bit for bit)")
    ap.add_argument("--warm-moves", type=int, default=2, help="untimed setup: full moves searched with the real evaluator before the "
                    "alignment steps, so that the timed window sees the evaluation cache as continuous self-play leaves it (restarted "
                    "games walking through openings that earlier games searched) instead of a cold table; 0 = none")
    ap.add_argument("--ring-ranks", choices=["auto", "all", "0"], default="auto", help="N>1: which ranks expand the gathered records into a dense "
                    "replay ring in their HBM: all (every rank may sample the union), 0 (rank 0 only: the trainer's rank); auto = 0 when a "
                    "trainer runs (--train-every), all otherwise")
    ap.add_argument("--dist-timeout", type=int, default=180, help="N>1: seconds a collective may wait for a peer before the job fails")
    ap.add_argument("--cache-verify", action="store_true", help="evaluation cache debug mode (CCZ_FLAG_CACHE_VERIFY): one table hit in 128 is "
                    "evaluated again and compared bit for bit; the count of mismatches is in the line")
    ap.add_argument("--inject-fault", default="", help="testing: RANK:STEP raises on that rank at that timed step (the fail-fast path)")
    ap.add_argument("--value-f16", action="store_true", help="accumulate Q in float16 as the reference's CUDA path does (CCZ_FLAG_VALUE_F16, "
                    "net.py:178-189 -> mcts.py:63-71); default: float32, its CPU path")
    return ap.parse_args()


def host_cores() -> int:
    """CPU threads this process may really use: affinity mask, cgroup quota, and the GPU box's 16-core share."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(seconds: float, blocks: int, channels: int):
    """Oracle leg: BASELINE config[0] restated -- ONE game, strictly sequential PUCT (oracle/xq_mcts.c,
    restating mcts.py), n_playout=200, batch-1 reference-architecture net on the host CPU in fp32 -- with the time split
    net / rules / tree BASELINE.md section 3 promised, and the same loop with a constant-time stub evaluator (the non-net part
    on its own). 80 % of the budget goes to the net leg, 20 % to the stub leg."""
    from oracle import OracleBoard, OracleMCTS
    from chinesechesszero_amd.net import Net

    torch.manual_seed(0)
    net = Net(channels, blocks).eval()
    cores = host_cores()
    torch.set_num_threads(cores)
    t_net = [0.0]
    t_enc = [0.0]

    def net_evaluator(board, ids):
        t0 = time.perf_counter()
        x = torch.from_numpy(board.leaf_planes()[None])   # decode_board + the 17 plane groups (tools.py:74-106, net.py:160-177)
        t1 = time.perf_counter()
        with torch.no_grad():
            logp, v = net(x)
        p = np.exp(logp.numpy().reshape(-1))
        t_enc[0] += t1 - t0
        t_net[0] += time.perf_counter() - t1
        return p[ids], v.numpy().reshape(-1)[0]

    def stub_evaluator(board, ids):
        return np.full(len(ids), 1.0 / 2086, np.float32), 0.0

    def leg(evaluator, budget):
        board = OracleBoard()
        mcts = OracleMCTS(evaluator, c_puct=5, n_playout=200)
        mcts.set_timing(True)
        rs = np.random.RandomState(0)
        sims = moves = 0
        t0 = time.perf_counter()
        deadline = t0 + budget
        while time.perf_counter() < deadline and not board.is_game_over():
            done_move = True
            for _ in range(200):
                mcts.playout(board)
                sims += 1
                if time.perf_counter() >= deadline:
                    done_move = False
                    break
            if not done_move:
                break
            acts, visits, _, _ = mcts.root_children()
            temp = 1.0
            x = 1.0 / temp * np.log(visits.astype(np.int64) + 1e-10)
            pr = np.exp(x - x.max())
            pr /= pr.sum()
            move = int(rs.choice(acts, p=0.75 * pr + 0.25 * rs.dirichlet(0.2 * np.ones(len(pr)))))
            mcts.update_with_move(move)
            board.push_id(move)
            moves += 1
        dt = time.perf_counter() - t0
        rules_s, tree_s = mcts.timers()
        return sims, moves, dt, rules_s, tree_s

    sims, moves, dt, rules_s, tree_s = leg(net_evaluator, 0.8 * seconds)
    rules_s += t_enc[0]                                   # the leaf encoding is rules-side work (tools.decode_board)
    other = max(0.0, dt - t_net[0] - rules_s - tree_s)    # Python glue: ctypes callback, array conversions, move choice
    s_sims, s_moves, s_dt, s_rules, s_tree = leg(stub_evaluator, 0.2 * seconds)
    return {"value": sims / dt, "unit": "sims/s", "cores": cores, "kind": "port",
            "sample": f"{sims} sequential playouts ({moves} full moves) of one self-play game, n_playout=200, "
                      f"batch-1 {blocks}x{channels} net fp32 on CPU, {dt:.1f} s; net share {t_net[0] / dt:.2f}",
            "moves_per_sec": moves / dt if moves else sims / dt / 200.0,
            # where a CPU playout spends its time (C rules and tree of the oracle: an OPTIMISTIC stand-in for
            # the reference's pure-Python cchess + mcts.py; the net is the reference's architecture on the same threads)
            "split": {"net": t_net[0] / dt, "rules": rules_s / dt, "tree": tree_s / dt, "python_glue": other / dt,
                      "ms_per_playout": {"net": 1e3 * t_net[0] / max(1, sims), "rules": 1e3 * rules_s / max(1, sims),
                                         "tree": 1e3 * tree_s / max(1, sims)}},
            "stub": {"value": s_sims / s_dt, "unit": "sims/s", "what": "the same sequential loop with a constant-time evaluator "
                     "(uniform priors, v = 0): rules + tree + Python callback glue only",
                     "sample": f"{s_sims} playouts ({s_moves} moves), {s_dt:.1f} s",
                     "split": {"rules": s_rules / s_dt, "tree": s_tree / s_dt, "python_glue": max(0.0, s_dt - s_rules - s_tree) / s_dt}}}


def preroll(e, plies: int, stagger: bool):
    """Untimed setup: bring the boards to a steady-state spread of game phases.

    Self-play in steady state has its boards at every phase of a game (finished boards restart at once), not all on the
    opening position. ``plies`` lockstep plies are played with the stub evaluator and ONE simulation per ply (root
    expansion; a flat pi at temperature 1e3 => a uniformly random legal move, what a random-init net plays up to noise);
    with ``stagger`` board b is restarted at ply (b mod plies), so that at the end the boards sit at plies 1..plies of
    their games, evenly. Games that end on the way are harvested (rows discarded) and restarted."""
    from chinesechesszero_amd.net import uniform_evaluator
    B = e.B
    temps = np.full(B, 1e3, np.float64)
    b_idx = np.arange(B)
    for t in range(plies):
        if stagger and t > 0:
            mask = ((b_idx % plies) == t).astype(np.uint8)
            if mask.any():
                e.reset(mask)
        leaf = e.select_leaves()
        e.expand_backup(*uniform_evaluator(leaf))
        e.finish_move(temps=temps, keep_tree=False)
        if e.game_status()["over"].any():
            for _ in e.harvest_chunks(1 << 16):
                pass
    e.check_healthy()


LIVE_OVER_PROFILE_BAND = (1.15, 1.4)   # this run's raw HIP-event k_step figure over the committed rocprofv3 average: what an event pair's overhead explains


def committed_profile(path: str, head: str, workload: dict):
    """The committed rocprofv3 summary (profiles/pmc_summary.json) IF it describes what this process runs: taken with the same
    code (``head`` == build.code_hash() of the run that was profiled) on the same workload. Returns ``(summary | None, why not |
    None, the profile's head)``. A profile of other code or another workload is refused, never replayed silently."""
    if not os.path.exists(path):
        return None, "no committed profile", None
    try:
        with open(path) as f:
            pm = json.load(f)
    except Exception as exc:
        return None, f"unreadable profile: {exc}", None
    ph = pm.get("head")
    wl = dict({"boards_per_gpu": 4096, "sims_per_move": 400, "evaluator": "net", "max_plies": 200}, **pm.get("workload", {}))
    full = {"blocks": 40, "channels": 256, "preroll_plies": 200, "align": True}   # what run_profile.sh's passes run with
    want = dict(full, **wl)
    diff = {k: (workload.get(k), v) for k, v in want.items() if workload.get(k) != v}
    if diff:
        return None, f"the committed profile is of another workload ({', '.join(f'{k}: {a} vs {b}' for k, (a, b) in sorted(diff.items()))})", ph
    if not ph:
        return None, "the committed profile does not say which code it was taken with (no head)", ph
    if ph != head:
        return None, f"the committed profile was taken with other code (head {ph}, this run {head})", ph
    return pm, None, ph


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from chinesechesszero_amd import launch   # (touches no GPU)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (a child process; nothing has touched the GPU yet). A node
        # with fewer GPUs than ranks is refused here, in one line, before any rank exists.
        raise SystemExit(launch.self_launch(__file__, a.gpus, sys.argv[1:], share_gpu=a.share_gpu, cores=host_cores()))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    solo_group = bool(a.rccl_group_of_one and world == 1)
    multi = world > 1 or solo_group   # the N>1 code path: process group, exchange, replay ring, multi_gpu block
    if a.graph and (a.evaluator != "stub" or world > 1 or a.train_every > 0):
        raise SystemExit("--graph is for the simulator-only line: --evaluator stub on one GPU without a trainer")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    if a.share_gpu:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: cuda:{local_rank} does not exist ({torch.cuda.device_count()} device(s) visible): --gpus {a.gpus} needs one GPU per rank")
    if world > 1:  # every rank runs MIOpen's find step: keep their user perf-db / kernel caches apart
        os.environ.setdefault("MIOPEN_USER_DB_PATH", f"/tmp/cczero_miopen_rank{rank}")
        os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", f"/tmp/cczero_miopen_rank{rank}/cache")
        os.makedirs(os.environ["MIOPEN_CUSTOM_CACHE_DIR"], exist_ok=True)
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if solo_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(launch.free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if multi:
        # 180 s, not torch's 10 minutes: a rank that dies mid-window must not keep its peers in the all-gather for long
        launch.init_distributed(a.backend, dev, timeout_s=a.dist_timeout)
    xdev = dev if a.backend == "nccl" else torch.device("cpu")  # where the exchange buffers live

    from chinesechesszero_amd.build import code_hash
    from chinesechesszero_amd.net import PolicyValueNet, uniform_evaluator
    from chinesechesszero_amd.replay import AsyncRecordExchange, RecordGatherer, ReplayBuffer, exchange_finished_games
    from chinesechesszero_amd.selfplay import BatchedSelfPlay

    n = a.playout
    if a.evaluator == "net":
        torch.manual_seed(0)
        pvn = PolicyValueNet(device=dev, num_channels=a.channels, resblocks_num=a.blocks)
        pvn.refresh_inference_copy()
        evaluator = pvn.evaluate_leaves_logits  # compact boundary: logits in, the engine gathers the legal priors
    else:
        evaluator = uniform_evaluator
    # the "shared replay buffer" of BASELINE configs[3]: the union of all ranks' rows as a dense ring in HBM; finished games
    # arrive as compact records and are expanded straight into the ring (ccz_expand_records) -- on the ranks that CONSUME it:
    # every rank without a trainer ("all": any rank may sample), rank 0 only when a trainer runs there (configs[4]: the other
    # ranks would write 8 x 28 k rows x 29,768 B = 6.8 GB per move into rings nobody reads)
    ring_ranks = a.ring_ranks if a.ring_ranks != "auto" else ("0" if a.train_every > 0 else "all")
    has_ring = (multi or a.train_every > 0) and (ring_ranks == "all" or rank == 0)
    rb = ReplayBuffer(a.replay_rows or 40000 * world, dev) if has_ring else None
    bad_records = torch.zeros(1, dtype=torch.int32, device=dev)

    trainer = train_once = None
    if a.train_every > 0 and rank == 0:
        from chinesechesszero_amd.trainer import Trainer
        torch.manual_seed(1)
        # bf16 autocast, no GradScaler: the fp16 scaler's step() reads found_inf on the host and would stall the self-play
        # launch loop on every update
        trainer = Trainer(PolicyValueNet(device=dev, num_channels=a.channels, resblocks_num=a.blocks), amp_dtype="bf16")
        # the buffer is prefilled with synthetic rows so that the trainer runs at its steady-state cadence from the first
        # step ("data": "synthetic"); harvested rows are appended as games finish
        g = torch.Generator(device=dev).manual_seed(2)
        ps = torch.rand((8192, 2086), device=dev, generator=g)
        rb.append((torch.rand((8192, 17, 7, 10, 9), device=dev, generator=g) > 0.9).half(), ps / ps.sum(1, keepdim=True),
                  torch.randint(-1, 2, (8192,), device=dev, generator=g).float())
        side = torch.cuda.Stream(device=dev)
        side_done = torch.cuda.Event()
        side_done.record(torch.cuda.current_stream(dev))
        train_steps = [0]

        def train_once():
            """One trainer update (batch 2048 sampled from the replay ring) on the side stream of GPU0, no host wait."""
            side.wait_stream(torch.cuda.current_stream(dev))  # replay-buffer appends (main stream) happen before the sample
            with torch.cuda.stream(side):
                trainer.step(*rb.sample(2048), sync=False)
                side_done.record(side)

    mul = 2   # dense rows per ply record (the sample and its mirror image, collect.py:112-131)
    max_plies_eff = a.max_plies if a.max_plies > 0 else 2048
    gather = ex = None
    if multi:
        if a.exchange == "sync":
            gather = RecordGatherer(max(a.gather_plies, max_plies_eff), xdev, always_collective=solo_group)
        else:
            ex = AsyncRecordExchange(max(a.gather_plies, max_plies_eff), xdev, timeout_s=a.dist_timeout, always_collective=solo_group)
    slow_rank, slow_s = (int(a.slow_rank.split(":")[0]), float(a.slow_rank.split(":")[1])) if a.slow_rank else (-1, 0.0)
    sp = e = None              # this rank's engine: made by make_engine() below, once the rank's board count is known
    calibrating = [False]      # rank 0 measuring itself for --boards-rank0 auto: finished games are dropped, not exchanged
    trainer_on = [True]

    def make_engine(boards, base):
        return BatchedSelfPlay(evaluator, boards, n_playout=n, seed=0, board_id_base=base, device=local_rank, use_graph=a.graph,
                               sampling="device", max_plies=a.max_plies, value_f16=a.value_f16,
                               eval_cache_log2=a.eval_cache_log2 if a.evaluator == "net" else 0, cache_verify=a.cache_verify)

    ev = lambda: torch.cuda.Event(enable_timing=True)
    boundary = {"events": [], "host_s": 0.0, "n": 0, "rows": 0, "rows_local": 0, "games": 0, "gather_s": 0.0, "collectives": 0, "expand_s": 0.0,
                "exchanges_completed": 0, "slept_s": 0.0}
    timing = [False]
    step_no = [0]

    def consume(done_list):
        """Completed exchanges (AsyncRecordExchange): the dense rows of ALL ranks' games rebuilt in this rank's replay ring
        (k_expand_records on the main stream, asynchronous) -- or only counted, on a rank that consumes no rows."""
        for x in done_list:
            g1 = time.perf_counter()
            if rb is not None:
                if trainer is not None:
                    torch.cuda.current_stream(dev).wait_event(side_done)  # the trainer's gather reads must not race the append
                rows = rb.append_records(x.union.to(dev, non_blocking=True), e.record_flags(), e.plane_of_type, bad=bad_records)
            else:
                rows = mul * int(x.union.shape[0])
            if timing[0]:
                boundary["rows"] += rows
                boundary["games"] += x.games
                boundary["exchanges_completed"] += 1
                boundary["expand_s"] += time.perf_counter() - g1

    def per_move():
        """The move boundary: pi + Dirichlet-mixed choice + re-root + push + game end (k_finish_move, k_flip_half), tuple
        harvest of the games that ended (k_harvest) with restart, and for N > 1 the all-gather of those rows."""
        timed = timing[0]
        if ex is not None and not calibrating[0]:
            # the launch loop runs ahead of the GPU and the boundary starts with a read-back that waits for everything queued: wait
            # HERE, ticking, so that an exchange the peers have announced is issued now and not when this host returns from its sync
            caught_up = torch.cuda.Event()
            caught_up.record(torch.cuda.current_stream(dev))
            while not caught_up.query():
                consume(ex.tick_until(caught_up))
        if timed:
            # the launch loop runs ahead of the GPU; the boundary's first host read would wait for the queued steps anyway:
            # drain them here so that the boundary's own wall time is what gets measured (no extra wait in total)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        if timed:
            m0, m1 = ev(), ev()
            m0.record()
        moves = sp.finish_move()
        st = e.game_status()
        done = int(st["over"].sum())
        if timed and rank == slow_rank:   # testing: this rank falls behind at its move boundary (--slow-rank)
            time.sleep(slow_s)
            boundary["slept_s"] += slow_s
        if calibrating[0]:
            rows = loc = 0
            for _ in (e.harvest_record_chunks(1 << 16) if done else ()):
                pass               # (a calibration run's games belong to no job: harvested so that the boards restart, then dropped)
        elif ex is not None:
            # no rank waits for another one here: the records join this rank's backlog, the rank announces the next exchange, and
            # whatever exchange has completed meanwhile is expanded into the ring (replay.AsyncRecordExchange)
            chunks = list(sp.harvest_record_chunks(ex.cap)) if done else []
            loc = mul * sum(int(c.shape[0]) for c in chunks)
            rows = 0
            consume(ex.post(chunks, games=done))
        elif gather is None:
            chunks = list(e.harvest_chunks(1 << 19)) if done else []
            rows = sum(int(c[2].shape[0]) for c in chunks)
            if trainer is not None and rows:
                torch.cuda.current_stream(dev).wait_event(side_done)  # the trainer's gather reads must not race the append
                for c in chunks:
                    rb.append(*c)
            loc = rows
        else:
            rows = loc = 0
            for union, games in exchange_finished_games(sp, gather, done):   # k_harvest_records + ONE collective per iteration
                g1 = time.perf_counter()
                if rb is not None:
                    if trainer is not None:
                        torch.cuda.current_stream(dev).wait_event(side_done)  # the trainer's gather reads must not race the append
                    # the dense rows of ALL ranks' games rebuilt in this rank's replay ring (k_expand_records, asynchronous)
                    rows += rb.append_records(union.to(dev, non_blocking=True), e.record_flags(), e.plane_of_type, bad=bad_records)
                else:
                    rows += mul * int(union.shape[0])   # received, not expanded: this rank consumes no rows
                if timed:
                    boundary["gather_s"] += gather.seconds
                    boundary["collectives"] += gather.collectives
                    boundary["games"] += games
                    boundary["expand_s"] += time.perf_counter() - g1
                loc += mul * gather.rows_per_rank[gather.rank]
        if timed:
            m1.record()
            torch.cuda.synchronize()
            boundary["events"].append((m0, m1))
            boundary["host_s"] += time.perf_counter() - t0
            boundary["n"] += 1
            boundary["rows"] += rows
            boundary["rows_local"] += loc
            if gather is None and ex is None:
                boundary["games"] += done
        return moves

    pairs, trace, cur = [], [], {}

    def hooks(stage, i):
        """HIP events around the evaluator side and the simulator kernel of every timed step (on the stream they are launched
        on), the step counter, the concurrent trainer's cadence and the fault injection of the fail-fast test."""
        if stage == "eval0":
            if timing[0]:
                cur["e"] = (ev(), ev(), ev())
                cur["e"][0].record()
            return
        if stage == "eval1":
            if timing[0]:
                cur["e"][1].record()
            return
        if timing[0]:
            e0, e1, e2 = cur["e"]
            e2.record()
            if i + 1 < n:
                pairs.append((e0, e1, e2))
            trace.append((i, e0, e2))
            if a.inject_fault and a.inject_fault == f"{rank}:{len(trace)}":
                raise RuntimeError(f"injected fault on rank {rank} at timed step {len(trace)} (--inject-fault)")
        step_no[0] += 1
        if ex is not None and not calibrating[0]:
            consume(ex.tick())   # a host-side check (nearly always nothing): issue the all-gather once every rank announced it, pick up a finished one
        if trainer is not None and trainer_on[0] and step_no[0] % a.train_every == 0:
            train_once()
            train_steps[0] += 1 if timing[0] else 0  # updates inside the timed window

    def run(steps, timed):
        """``steps`` lockstep simulations through the PRODUCT's loop (BatchedSelfPlay.advance: evaluator -> fused k_step; a move
        boundary flushes with expand_backup, plays the move, harvests and re-selects)."""
        timing[0] = timed
        sp.advance(steps, hooks=None if a.graph else hooks, boundary=per_move)   # (--graph: the captured graph is replayed only without hooks)
        timing[0] = False

    def timed_steps(k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(k, False)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k

    def board_states_and_trees():
        """Steady-state board states (preroll) and, with the real evaluator, whole searched moves (trees, games, evaluation cache)."""
        if a.preroll_plies > 0:
            preroll(e, a.preroll_plies, stagger=True)
        if a.evaluator == "net":
            run(a.warm_moves * n, False)   # whole moves, boundaries included: trees, games and the evaluation cache in steady state

    # ---- boards per rank. All ranks hold --boards boards except, on request, rank 0 (the rank that shares its GPU with the trainer).
    # ``auto``: rank 0 measures ITSELF on the real workload -- a full engine of --boards boards brought to the steady state, then a
    # few trainer periods of real steps without and with the trainer -- and takes the board count at which its step NEXT TO the
    # trainer costs what a plain rank's step costs (the trainer adds a constant per step: measured, profiles/r05_*). The full engine
    # is then dropped and rank 0 builds the one it runs with. Global board ids are a prefix sum over the final counts.
    t_setup = time.perf_counter()
    boards_rank0, calibration = None, None
    if a.boards_rank0 == "auto":
        if a.train_every <= 0 or a.evaluator != "net":
            raise SystemExit("--boards-rank0 auto balances rank 0 against its trainer: it needs --train-every and the real evaluator")
        t = torch.zeros(1, dtype=torch.int64, device=xdev)
        if rank == 0:
            sp = make_engine(a.boards, 0)
            e = sp.engine
            calibrating[0] = True
            board_states_and_trees()
            periods = 3
            trainer_on[0] = False
            t_plain = timed_steps(periods * a.train_every)
            trainer_on[0] = True
            train_once()
            t_with = timed_steps(periods * a.train_every)
            calibrating[0] = False
            c = max(0.0, t_with - t_plain)                       # what the trainer adds to a step, whatever the board count
            b0 = a.boards * max(0.0, t_plain - c) / t_plain
            q = 64
            b0q = int(max(q, min(a.boards, q * int(b0 // q))))
            calibration = {"boards_rank0": b0q, "plain_step_ms": 1e3 * t_plain, "step_ms_with_trainer_at_full_boards": 1e3 * t_with,
                           "trainer_ms_per_step": 1e3 * c, "trainer_bound": b0 < q,
                           "what": f"rank 0, real workload ({a.boards} boards in steady state, {a.warm_moves} searched moves): {periods * a.train_every} steps without "
                                   f"and {periods * a.train_every} with the trainer (one update of batch 2048 per {a.train_every} steps, side stream); boards_rank0 = "
                                   "boards x (plain - trainer) / plain, rounded down to 64 (at least 64: trainer_bound says when even that is too many)"}
            t[0] = b0q
            if b0q != a.boards:
                e.close()
                sp = e = None
                torch.cuda.empty_cache()
        if multi:
            dist.broadcast(t, src=0)
        boards_rank0 = int(t.item())
    elif a.boards_rank0:
        boards_rank0 = int(a.boards_rank0)
    counts, bases = launch.board_partition(world, a.boards, boards_rank0)
    B = counts[rank]

    # ---- untimed setup: steady-state board states, trees warmed with the REAL evaluator up to the point where the
    # timed window starts, so that the K timed steps straddle a real move boundary of every board (the end of one
    # move's search, k_finish_move + harvest + restart [+ all-gather], the start of the next move on the kept subtree)
    if sp is None:
        sp = make_engine(B, bases[rank])
        e = sp.engine
        board_states_and_trees()
    st_pre = e.game_status()
    half = min(a.steps, n) // 2
    phase = (n - half - a.warmup) % n if a.align else 0
    to_go = (phase - sp._sim) % n             # (a calibration engine that was kept stands a few steps into its move)
    if a.align_evaluator == "stub" and a.evaluator == "net":
        real = (sp.evaluator, sp.planned)     # (the pending leaf does not care which evaluator answers it)
        sp.evaluator, sp.planned = uniform_evaluator, False
        run(to_go, False)
        sp.evaluator, sp.planned = real
    else:
        if calibration is not None and to_go >= 2 * a.train_every:
            calibration["step_ms_with_trainer_at_boards_rank0"] = 1e3 * timed_steps(to_go)   # the balance reached, on the engine that runs
        else:
            run(to_go, False)
    assert sp._sim == phase % n
    setup_s = time.perf_counter() - t_setup

    if gather is not None:  # one untimed exchange: communicator / channel set-up of the collective is not part of a move
        gather.gather(torch.empty((0, 880), dtype=torch.uint8, device=xdev))
    run(a.warmup, False)
    if ex is not None:      # drain what the untimed moves left (and set the communicator up): nothing is in flight at the window's barrier
        for x in ex.flush_iter():
            consume([x])
        ex_stats0 = (ex.collectives, ex.host_seconds, ex.plies_sent, ex.issued, ex.bytes_sent)
        ex.max_call_s = 0.0
    torch.cuda.synchronize()
    s0 = e.stats()
    tower_probe = None
    if a.evaluator == "net":
        tower_probe = pvn._infer.tower_probe = []   # one HIP-event pair around the tower's 2 x blocks launches of every timed step
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.steps, True)
    if ex is not None:   # (waiting for the queued steps: keep the exchange moving meanwhile)
        caught_up = torch.cuda.Event()
        caught_up.record(torch.cuda.current_stream(dev))
        timing[0] = True
        while not caught_up.query():
            consume(ex.tick_until(caught_up))
        timing[0] = False
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0  # this rank's own time (the per-rank rates); the job's time is taken behind the barrier
    drain_s = 0.0
    if ex is not None:
        # INSIDE the job's timed region: every record harvested in the window is delivered to every rank before the clock stops
        # (the drain waits for the slowest rank, as the barrier behind it would anyway)
        d0 = time.perf_counter()
        ex_window = (ex.collectives - ex_stats0[0], ex.host_seconds - ex_stats0[1], ex.plies_sent - ex_stats0[2], ex.max_call_s, ex.issued - ex_stats0[3])
        timing[0] = True
        for x in ex.flush_iter():
            consume([x])
        timing[0] = False
        torch.cuda.synchronize()
        drain_s = time.perf_counter() - d0
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if a.evaluator == "net":
        pvn._infer.tower_probe = None
    s1 = e.stats()
    ranks_seen, per_rank, err_any, bad_total = 1, None, int(s1["error_flags"]), int(bad_records.item())
    dev_all = [s1["pruned_subtrees"] - s0["pruned_subtrees"], s1["truncated_games"] - s0["truncated_games"], s1["pruned_subtrees"], s1["truncated_games"]]
    if multi:
        t = torch.zeros(3 * world + 8, dtype=torch.float64, device=xdev)
        t[0] = dt
        mx = t[:1].clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        t[0] = 0.0
        t[1 + rank] = B * a.steps / dt_local
        t[world + 1] = 1.0                                      # ranks that got here
        t[world + 2] = float(bad_total)                         # records of cut games seen by the expansion, all ranks
        t[world + 3] = 1.0 if s1["error_flags"] else 0.0        # ranks whose engine raised a sticky error bit
        t[world + 4 + rank] = 1e3 * dt_local / a.steps          # this rank's own step time (its K steps, boundary included, no waiting for peers)
        t[2 * world + 4 + rank] = 1e3 * (ex.max_call_s if ex is not None else 0.0)   # the longest single call into the exchange (host)
        # where the engines departed from the reference, summed over the ranks (the line's `deviations`)
        t[3 * world + 4] = float(s1["pruned_subtrees"] - s0["pruned_subtrees"])
        t[3 * world + 5] = float(s1["truncated_games"] - s0["truncated_games"])
        t[3 * world + 6] = float(s1["pruned_subtrees"])
        t[3 * world + 7] = float(s1["truncated_games"])
        dist.all_reduce(t)
        dt = float(mx.item())
        ranks_seen = int(t[world + 1].item())
        bad_total = int(t[world + 2].item())
        err_any = int(t[world + 3].item())
        per_rank = [float(v) for v in t[1:world + 1].tolist()]
        rank_step_ms = [float(v) for v in t[world + 4:2 * world + 4].tolist()]
        rank_xchg_max_call_ms = [float(v) for v in t[2 * world + 4:3 * world + 4].tolist()]
        dev_all = [int(v) for v in t[3 * world + 4:3 * world + 8].tolist()]
    e.check_healthy()
    st_end = e.game_status()
    sims = s1["sims"] - s0["sims"]
    probes = s1["cache_probes"] - s0["cache_probes"]
    planned = sp.planned
    rows_per_step = ((probes - (s1["cache_hits"] - s0["cache_hits"]) - (s1["cache_shared_rows"] - s0["cache_shared_rows"])) / a.steps) if planned else float(B)

    # the evaluator's dominant kernel (2 x blocks launches per step): the fused tower convolution. In the WINDOW: the HIP-event pair
    # around the tower of every timed step over the rows those steps really computed; after it: a full-batch tower on the
    # activations of a real leaf batch (what round 3 reported as the only figure)
    net_roofline = None
    if a.evaluator == "net" and rank == 0 and a.channels == 256 and B >= 192 and pvn._infer.opt_fused_conv():
        net_roofline = tower_roofline(a, pvn, e, sp, B, ev, tower_probe, rows_per_step, 1e3 * dt / a.steps)

    if os.environ.get("CCZ_BENCH_TRACE") and rank == 0:  # per-step GPU time inside the timed window (diagnostics, stderr)
        print("trace: (simulation index within its move, ms) " + " ".join(f"{i}:{x.elapsed_time(y):.2f}" for i, x, y in trace[:4000]), file=sys.stderr)
    exp = max(1, s1["expansions"] - s0["expansions"])
    kbar = (s1["sum_children"] - s0["sum_children"]) / exp
    dbar = (s1["sum_depth"] - s0["sum_depth"]) / max(1, sims)
    out = None
    if rank == 0:
        total_boards = sum(counts)
        value = total_boards * a.steps / dt
        if pairs:
            # MEDIAN of the per-step HIP-event pairs (a stray long interval -- a boundary's neighbour, a trainer burst -- must not move it)
            t_net = float(np.median([p[0].elapsed_time(p[1]) for p in pairs])) * 1e-3
            t_step = float(np.median([p[1].elapsed_time(p[2]) for p in pairs])) * 1e-3
        else:
            t_net = t_step = float("nan")
        graph_step = None
        if a.graph:   # no per-step events: the whole step by the host clock, boundary share removed (graph launch + k_step: an upper bound of k_step)
            graph_step = t_step = (dt - boundary["host_s"]) / a.steps
        # algorithmic bytes per simulation, SURVEY 8(d): the fused k_step kernel does all of it. The
        # 21,420-B evaluator input is counted at the 3,780 B that can be non-zero (groups 7/15/16);
        # the 14 static-zero groups are written once at create, not per simulation.
        a_sel = 3780 + 12 * kbar * dbar + 6 * dbar + 180 + 2 * kbar
        a_exp = (4 * kbar + 4) + 18 * kbar + 16 * (dbar + 1)
        a_step = a_sel + a_exp
        a_sim_survey = a_step - 3780 + 21420
        # roofline.frac / achieved = THIS run's algorithmic bytes over k_step's average launch duration MEASURED LIVE by this run: the
        # median of the HIP-event pairs around every k_step launch of the window, on its stream. An event pair adds ~9 us to a ~33 us
        # kernel and nothing is subtracted (round 4 subtracted a separately measured "floor" and over-corrected by 6 us): the figure is
        # a LOWER bound of the fraction, named as one. (Round 5 put the committed rocprofv3 average into the headline when the code
        # hash matched: a slower or throttled box then still reported the profiled box's fraction -- ADVICE r05.) The committed
        # profile's duration is quoted under its OWN keys, only when it is of this code and workload AND this run's live figure sits
        # where an event pair's overhead puts it (1.15 .. 1.4 x the profile's): otherwise k_step, the box or the clock differ.
        head = code_hash()
        ach_raw = a_step * B / t_step if t_step == t_step else 0.0
        traffic = traffic_source = rocprof_ns = pmc_window = None
        wl_now = {"boards_per_gpu": B, "sims_per_move": n, "evaluator": a.evaluator, "max_plies": a.max_plies, "blocks": a.blocks,
                  "channels": a.channels, "preroll_plies": a.preroll_plies, "align": bool(a.align)}
        pm, profile_why, profile_head = committed_profile(os.path.join(ROOT, "profiles", "pmc_summary.json"), head, wl_now)
        if pm is not None:
            ks = pm.get("k_step", {})
            rocprof_ns = ks.get("avg_ns")
            pmc_window = ks.get("window")
            traffic = (pmc_window or {}).get("hbm_bytes_per_launch", ks.get("hbm_bytes_per_launch"))
            traffic_source = f"profiles/pmc_summary.json ({pm.get('run', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, separate runs')}), head {profile_head}: replayed, not live"
            if net_roofline is not None:  # HBM bytes of one convolution launch (PMC passes of profiles/run_profile.sh)
                per_kernel = pm.get("k_conv3x3", {}).get("hbm_bytes_per_launch")
                # one layer over the live rows = groups x chains launches of the tower kernel(s), two per chain with the edge-pair
                # kernel; the committed counter is the average over ALL k_conv3x3* launches (middle and edge alike)
                net_roofline["traffic"] = per_kernel * net_roofline["chains"] * net_roofline["groups"] * net_roofline["kernel_launches_per_chain_and_layer"] if per_kernel else None
                net_roofline["traffic_source"] = traffic_source
        t_used, dur_src = (t_step if t_step == t_step else None), ("the whole step by the host clock under hipGraph replay, move boundaries removed (graph launch + k_step: an upper "
                                                                   "bound of k_step's duration, a lower bound of the fraction)" if graph_step is not None else
                                                                   "LIVE: RAW HIP events around every k_step launch of the timed window on its stream (median; "
                                                                   "the ~9 us an event pair adds are NOT subtracted: a lower bound of the fraction)")
        ach = a_step * B / t_used if t_used else 0.0
        live_vs_profile = (t_step / (rocprof_ns * 1e-9)) if (rocprof_ns and t_step == t_step and graph_step is None) else None
        frac_profile = us_profile = None
        if rocprof_ns and live_vs_profile is not None and LIVE_OVER_PROFILE_BAND[0] <= live_vs_profile <= LIVE_OVER_PROFILE_BAND[1]:
            us_profile, frac_profile = rocprof_ns * 1e-3, a_step * B / (rocprof_ns * 1e-9) / HBM_PEAK
            profile_note = (f"rocprofv3 --kernel-trace --stats average of k_step in profiles/{pm.get('tag', 'rNN')}_kernel_stats.csv, taken with this code "
                            f"(head {head}) on this workload; this run's live figure is {live_vs_profile:.2f} x that (an event pair's own cost)")
        elif rocprof_ns and live_vs_profile is not None:
            profile_note = (f"the committed profile is of this code and workload, but this run's live k_step figure is {live_vs_profile:.2f} x its duration "
                            f"(expected {LIVE_OVER_PROFILE_BAND[0]} .. {LIVE_OVER_PROFILE_BAND[1]}): another box, clock or tree shape -- not quoted")
        else:
            profile_note = profile_why or ("no per-launch events under hipGraph replay" if graph_step is not None else None)
        # the move boundary, measured: HIP events around finish_move + harvest/restart (+ exchange) and the host wall
        # around the same region (the harvest and the exchange read counts on the host)
        mb_ev = float(np.mean([x.elapsed_time(y) for x, y in boundary["events"]])) if boundary["events"] else None
        mb_host = 1e3 * boundary["host_s"] / boundary["n"] if boundary["n"] else None
        step_ms = 1e3 * (dt - boundary["host_s"]) / a.steps  # one simulation step without the boundary share
        moves_per_sec = total_boards / ((n * step_ms + (mb_host or 0.0)) * 1e-3)
        flops = 8.551e9 * (a.blocks / 40.0) * (a.channels / 256.0) ** 2
        plies0, plies1 = st_pre["plies"], st_end["plies"]
        net_desc = f"random-init {a.blocks}x{a.channels} policy-value net fp16" if a.evaluator == "net" else "stub evaluator (uniform priors, v=0)"
        state_desc = (f"boards in steady state (plies 1..{a.preroll_plies} of their games, evenly; games adjudicated at {a.max_plies} plies)"
                      if a.preroll_plies > 0 else "all boards from the opening position")
        if a.evaluator == "net":
            state_desc += f", {a.warm_moves} untimed moves searched before the window" + (f" (they warm the evaluation cache of 2^{e.eval_cache_log2} positions)" if planned else "")
        out = {
            "metric": "self-play MCTS simulations/sec", "value": value, "unit": "sims/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # the reference has two arithmetic paths: CPU = float32 net + float32 Q, CUDA = fp16 autocast net + float16 Q
            # (net.py:178-199 -> mcts.py:63-71 under NEP 50). The headline pairs its GPU net precision with its CPU-path Q (what
            # the golden traces pin); --value-f16 is its CUDA path end to end (profiles/r03_bench_value_f16.json: the same rate)
            "dtype": ("u8 rules / f16 Q (reference CUDA path) / f64 PUCT (net: fp16)" if a.value_f16 else
                      "u8 rules / f32 Q (reference CPU path) / f64 PUCT (net: fp16)"), "data": "synthetic",
            "config": {"workload": f"{a.boards if world > 1 else B} concurrent boards/GPU x {n} sims/move, Dirichlet root noise on (device Philox), " + net_desc + ", " + state_desc,
                       "boards_per_gpu": a.boards if world > 1 else B, "boards_per_rank": counts, "sims_per_move": n, "evaluator": a.evaluator,
                       "preroll_plies": a.preroll_plies, "max_plies": a.max_plies, "warm_moves": a.warm_moves if a.evaluator == "net" else 0,
                       "window": f"{a.steps} steps starting at simulation {phase + a.warmup} of a move: "
                                 f"{boundary['n']} move boundary(ies) inside the timed window"},
            "moves_per_sec": moves_per_sec,
            "move_boundary": {"in_window": boundary["n"], "ms_events": mb_ev, "ms_host": mb_host,
                              "games_finished": boundary["games"], "rows_harvested_rank0": boundary["rows_local"],
                              "what": "k_finish_move + k_flip_half + status readback + " + ("k_harvest + restart" if not multi else
                                      "k_harvest_records + restart + all-gather of the records + k_expand_records into the replay ring"),
                              "moves_per_sec_formula": "n_gpus * boards / (sims_per_move * (ms_per_step without the boundary) + ms_host)"},
            "roofline": {"bound": "hbm", "kernel": "k_step (fused expand+backup+select+movegen+encode)",
                         "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach / HBM_PEAK,
                         "duration_source": dur_src, "avg_launch_us": (t_used or 0) * 1e6,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": a_step * B, "k_bar": kbar, "d_bar": dbar,
                         # (kept for readers of earlier rounds' lines: the same numbers as frac / avg_launch_us without --graph)
                         "avg_launch_us_hip_events_raw": (t_step * 1e6 if (t_step == t_step and graph_step is None) else None),
                         "frac_hip_events_raw": (ach_raw / HBM_PEAK if graph_step is None else None),
                         # the committed rocprofv3 profile's duration for k_step, under its own name (None + the reason when it does not apply)
                         "frac_at_committed_rocprofv3_duration": frac_profile, "avg_launch_us_committed_rocprofv3": us_profile,
                         "committed_profile_note": profile_note,
                         "code_hash": head, "profile_head": profile_head,
                         "live_over_profile": live_vs_profile,
                         # counters and algorithmic bytes of ONE pass (the PMC passes' own timed window): reproducible from profiles/
                         "pmc_window": pmc_window},
            "survey_a_sim_bytes": a_sim_survey,
            "step_split_us": {"k_step": (t_step * 1e6 if t_step == t_step else None), "evaluator": (t_net * 1e6 if t_net == t_net else None)},
            "trainer_updates": (train_steps[0] if trainer is not None else 0),
            "net_roofline": net_roofline,
            # rows the network really computed per step (with the evaluation cache: fewer than B) x 8.551 GFLOP
            "net_tflops": (flops * rows_per_step / t_net / 1e12) if (a.evaluator == "net" and t_net == t_net) else None,
            "eval_cache": ({"entries_log2": e.eval_cache_log2, "bytes": 528 << e.eval_cache_log2,
                            "leaves_needing_the_net_in_window": probes, "hits_in_window": s1["cache_hits"] - s0["cache_hits"],
                            "served_by_another_boards_row_in_window": s1["cache_shared_rows"] - s0["cache_shared_rows"],
                            "rows_computed_per_step": rows_per_step,
                            "fraction_of_needed_evaluations_skipped": 1.0 - rows_per_step * a.steps / max(1, probes),
                            "stores_total": s1["cache_stores"],
                            "verify": ({"hits_evaluated_again": s1["cache_verified"], "mismatches": s1["cache_verify_mismatches"]} if a.cache_verify else None),
                            "what": "positions evaluated before (this board / another board / another board of the same step) skip the network; "
                                    "the same trees bit for bit (tests/test_gpu_timed_path.py: cache on vs off at 4096 boards, oracle mirror through the planned boundary)"} if planned else None),
            # Where this engine departs from the reference, COUNTED (VERDICT r05 task 3; strict mode -- CCZ_FLAG_STRICT, what MCTS_AI and
            # the parity tests run with -- turns the first two into error bits): kept subtrees pruned at re-root time to fit the node
            # pool (the reference's tree is unbounded, mcts.py:31-39); games adjudicated as draws at max_plies (its game loop has no
            # cap, game.py:155; the workload string says what max_plies is here and why); table hits of the evaluation cache that were
            # evaluated again and disagreed (--cache-verify; None without it)
            "deviations": {"pruned_subtrees_in_window": dev_all[0], "pruned_subtrees_total": dev_all[2],
                           "truncated_games_in_window": dev_all[1], "truncated_games_total": dev_all[3],
                           "scope": "summed over all ranks" if multi else "this GPU",
                           "games_finished_total_rank0": s1["games"], "max_plies": a.max_plies,
                           "cache_verify_mismatches": (s1["cache_verify_mismatches"] if (planned and a.cache_verify) else None),
                           "cache_verify_hits_evaluated_again": (s1["cache_verified"] if (planned and a.cache_verify) else None),
                           "node_pool": {"nodes_peak": s1["nodes_peak"], "what": "pruning starts when a kept subtree exceeds cap - reserve nodes (ccz_config.max_nodes / reserve_nodes)"}},
            "engine_hbm_gb": s1["hbm_bytes"] / 1e9, "nodes_peak": s1["nodes_peak"], "depth_peak": s1["depth_peak"],
            "error_flags_any": err_any,
            "plies": {"start_mean": float(plies0.mean()), "start_max": int(plies0.max()), "end_mean": float(plies1.mean())},
            "setup_seconds": setup_s,
            "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),   # (chinesechesszero_amd/__init__.py: the tower's launch chains need queues of their own)
        }
        if multi:
            xg = gather if gather is not None else ex
            out["multi_gpu"] = {"world_size": world, "ranks_seen": ranks_seen, "backend": a.backend,
                                "error_flags_any": err_any, "bad_records": bad_total,
                                "boards_per_rank": counts, "board_id_base_per_rank": bases,
                                "per_rank_sims_per_sec": per_rank,
                                # every rank's own time per step over its K steps, boundary included, WITHOUT any wait for a peer
                                # (with --exchange async nothing in the window waits; the job's ms_per_step is the slowest rank + the drain)
                                "rank_step_ms": rank_step_ms,
                                "exchange": a.exchange, "exchanges_in_window": boundary["n"],
                                "rows_gathered": boundary["rows"],
                                "expand_ms_host": 1e3 * boundary["expand_s"] / max(1, boundary["n"]),
                                "wire_format": "compact ply records, 880 B per ply = 2 dense rows of 29,768 B (ccz_harvest_records -> "
                                               "all_gather_into_tensor -> ccz_expand_records into the replay ring of the ranks that consume rows)",
                                "bytes_sent_per_rank_per_collective": xg.bytes_per_exchange(), "gather_capacity_plies": xg.cap,
                                "payload_bytes_rank0_per_exchange": 880 * boundary["rows_local"] // mul // max(1, boundary["n"]),
                                "ring_ranks": ring_ranks, "replay_ring_rows": rb.cap if rb is not None else 0,
                                "replay_rows_total": rb.total if rb is not None else 0,
                                "dist_timeout_s": a.dist_timeout}
            if ex is not None:
                out["multi_gpu"].update({
                    # collectives ISSUED between the window's start and the end of its K steps, and by the drain behind them
                    "collectives_in_window": ex_window[0], "collectives_in_drain": ex.collectives - ex_stats0[0] - ex_window[0],
                    # exchanges DECIDED (window + drain), and how many of them were virtual: every rank had announced "nothing to send",
                    # so no collective was issued (the drain's closing exchange is normally one of these)
                    "exchanges_decided": ex.issued - ex_stats0[3], "exchanges_without_a_collective": ex.issued - ex_stats0[3] - (ex.collectives - ex_stats0[0]),
                    "bytes_sent_rank0": ex.bytes_sent - ex_stats0[4],
                    "exchanges_completed_in_window_and_drain": boundary["exchanges_completed"],
                    "games_gathered": boundary["games"],
                    # what rank 0's launch loop spent inside the exchange during its K steps (host seconds: announcements, store
                    # polls, issuing the collective, reading headers) and the longest single call of any rank
                    "exchange_host_ms_rank0": 1e3 * ex_window[1], "exchange_max_call_ms_per_rank": rank_xchg_max_call_ms,
                    "plies_sent_rank0": ex.plies_sent - ex_stats0[2], "backlog_peak_plies_rank0": ex.max_backlog_plies,
                    "drain_ms_rank0": 1e3 * drain_s,
                    "what": "replay.AsyncRecordExchange: post at the move boundary (backlog; staged + announced on the job's TCPStore: plies, flags), "
                            "ONE async all_gather_into_tensor of the smallest power-of-two slot that holds the largest announcement, from a side "
                            "stream, once every rank has announced; picked up by a later step's tick; an exchange in which nobody sends is decided "
                            "from the announcements alone. The drain (every record of the window delivered to every rank) is inside the timed region"})
            else:
                out["multi_gpu"].update({"collectives_in_window": boundary["collectives"],
                                         "gather_ms": 1e3 * boundary["gather_s"] / max(1, boundary["n"])})
            if solo_group:
                out["multi_gpu"]["rccl_group_of_one"] = ("one rank on backend nccl (= RCCL): the exchange path run for real on one GPU -- async all-gather from the "
                                                         "side stream, store handshake, expansion into the replay ring -- next to the full-size search")
            if slow_rank >= 0:
                out["multi_gpu"]["slow_rank"] = {"rank": slow_rank, "sleep_s_per_boundary": slow_s}
        if calibration is not None:
            out["rank0_calibration"] = calibration
        if world == 1 and not solo_group and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_baseline_seconds, a.blocks, a.channels)
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def tower_roofline(a, pvn, e, sp, B, ev, tower_probe, rows_per_step, ms_per_step):
    """MFMA roofline of the tower convolution (2 x blocks launches per step).

    ``frac`` / ``achieved`` / ``avg_launch_us`` describe the TIMED WINDOW: the HIP-event pair around the tower of every timed
    step (on the stream its launch chains fork from and join), divided by its 2 x blocks layers, against the algorithmic flops of
    the rows those steps really computed (the evaluation cache and terminal leaves remove rows). ``*_full_batch`` is the
    post-window figure round 3 printed: the whole 4096-row tower on the activations of a real leaf batch."""
    inf = pvn._infer
    layers = 2 * a.blocks
    flops_per_row_layer = 2.0 * 90 * 256 * 256 * 9           # 106.17 MFLOP: one 3x3 256->256 layer on one board
    g16 = inf._g16(B)
    groups = inf.tower_groups(B, g16)
    edge = inf._edge(B, g16)
    chains = inf.tower_chains(B, groups, edge)
    nr = {"bound": "mfma", "kernel": (("k_conv3x3_g16 (ranks 1..8) + k_conv3x3_g16_edge (ranks 0 / 9 of two groups per tile)" if edge else "k_conv3x3_g16") if g16 else "k_conv3x3_c256")
                                     + " (tower conv3x3 256->256 + bias + residual + ReLU, fp16 in / fp32 acc)",
          "row_layout": "group-of-16 (whole-rank tiles, off-board taps skipped)" if g16 else "nhwc (256-pixel tiles)",
          "peak": MFMA_PEAK_F16 / 1e12, "unit": "TFLOP/s", "traffic": None, "launches_per_step": layers, "groups": groups, "chains": chains,
          "kernel_launches_per_chain_and_layer": 2 if edge else 1,
          "note": "a 'launch' is one layer over the step's live rows, issued as groups x chains kernel launches over board ranges "
                  "(groups one after the other, the chains of a group concurrently)"}
    if tower_probe:
        t_layer = float(np.mean([x.elapsed_time(y) for x, y in tower_probe])) * 1e-3 / layers
        fl = flops_per_row_layer * rows_per_step
        nr.update({"achieved": fl / t_layer / 1e12, "frac": fl / t_layer / MFMA_PEAK_F16, "avg_launch_us": t_layer * 1e6,
                   "algorithmic_flops_per_launch": fl, "rows_per_launch": rows_per_step,
                   "duration_source": f"HIP events around the tower of each of the {len(tower_probe)} timed steps, / {layers} layers (live, in the window)",
                   # the algorithmic count is the convention for a padded 3x3 convolution (9 taps for every pixel); the group-of-16
                   # kernel does not issue the MFMAs of taps with dx off the board (150 of 162 per pair of ranks)
                   "mfma_flops_issued_per_launch": fl * ((150.0 / 162.0 if not edge else 150.0 / 162.0 * (1.0 - 0.2 / 3.0)) if g16 else 1.0)})
        # a figure that cannot fit in the step it describes must not be printed
        assert layers * nr["avg_launch_us"] <= ms_per_step * 1e3 * 1.001, (layers * nr["avg_launch_us"], ms_per_step)
    # post-window: the whole batch through the tower (no plan), three times on real activations
    leaf = sp._leaf if sp._leaf is not None else e.select_leaves()
    with torch.no_grad():
        x0 = inf._stem_fused(leaf)
        inf._tower_fused(x0.clone(memory_format=torch.preserve_format))
        xs = [x0.clone(memory_format=torch.preserve_format) for _ in range(3)]
        c0, c1 = ev(), ev()
        c0.record()
        for xi in xs:
            inf._tower_fused(xi)
        c1.record()
    torch.cuda.synchronize()
    t_conv = c0.elapsed_time(c1) * 1e-3 / (len(xs) * layers)
    nr.update({"avg_launch_us_full_batch": t_conv * 1e6, "frac_full_batch": flops_per_row_layer * B / t_conv / MFMA_PEAK_F16,
               "full_batch_note": f"after the window: all {B} rows through the tower, 3 passes; NOT the timed kernel (the window computed {rows_per_step:.0f} rows per step)"})
    if "frac" not in nr:   # no probe (path without the fused tower): the post-window figure is all there is
        nr.update({"achieved": flops_per_row_layer * B / t_conv / 1e12, "frac": nr["frac_full_batch"], "avg_launch_us": t_conv * 1e6,
                   "algorithmic_flops_per_launch": flops_per_row_layer * B, "duration_source": "post-window full batch"})
    return nr


if __name__ == "__main__":
    from chinesechesszero_amd.launch import guarded
    sys.exit(guarded(main))
