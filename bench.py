#!/usr/bin/env python3
"""bench.py -- self-play MCTS throughput of the MI355X lockstep engine (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one lockstep simulation of every board: select+make-move+movegen+encode (HIP) ->
policy-value net (PyTorch-ROCm fp16, same stream) -> expand+backup (HIP); every ``n_playout``-th
step also plays one move on every board (pi, Dirichlet-mixed choice, re-root, game end; HIP) and
exchanges finished training rows (RCCL all-gather when N > 1). Workload = BASELINE.json configs[2]:
4096 concurrent boards per GPU x 400 sims/move, Dirichlet root noise on, random-init 40x256 net,
all boards from the opening position (synthetic). Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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
    restating mcts.py), n_playout=200, batch-1 reference-architecture net on the host CPU in fp32."""
    import oracle
    from oracle import OracleBoard, OracleMCTS
    from chinesechesszero_amd.net import Net

    torch.manual_seed(0)
    net = Net(channels, blocks).eval()
    cores = host_cores()
    torch.set_num_threads(cores)
    t_net = [0.0]

    def evaluator(board, ids):
        x = torch.from_numpy(board.leaf_planes()[None])
        t0 = time.perf_counter()
        with torch.no_grad():
            logp, v = net(x)
        t_net[0] += time.perf_counter() - t0
        p = np.exp(logp.numpy().reshape(-1))
        return p[ids], v.numpy().reshape(-1)[0]

    board = OracleBoard()
    mcts = OracleMCTS(evaluator, c_puct=5, n_playout=200)
    rs = np.random.RandomState(0)
    sims = moves = 0
    t0 = time.perf_counter()
    deadline = t0 + seconds
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
    return {"value": sims / dt, "unit": "sims/s", "cores": cores, "kind": "port",
            "sample": f"{sims} sequential playouts ({moves} full moves) of one self-play game, n_playout=200, "
                      f"batch-1 {blocks}x{channels} net fp32 on CPU, {dt:.1f} s; net share {t_net[0] / dt:.2f}",
            "moves_per_sec": moves / dt if moves else sims / dt / 200.0}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with python -m torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    if world > 1:  # every rank runs MIOpen's find step: keep their user perf-db / kernel caches apart
        os.environ.setdefault("MIOPEN_USER_DB_PATH", f"/tmp/cczero_miopen_rank{rank}")
        os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", f"/tmp/cczero_miopen_rank{rank}/cache")
        os.makedirs(os.environ["MIOPEN_CUSTOM_CACHE_DIR"], exist_ok=True)
    import torch.distributed as dist
    if a.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)
    xdev = dev if a.backend == "nccl" else torch.device("cpu")  # where the exchange buffers live

    from chinesechesszero_amd.net import PolicyValueNet, uniform_evaluator
    from chinesechesszero_amd.replay import TupleGatherer
    from chinesechesszero_amd.selfplay import BatchedSelfPlay

    B, n = a.boards, a.playout
    if a.evaluator == "net":
        torch.manual_seed(0)
        pvn = PolicyValueNet(device=dev, num_channels=a.channels, resblocks_num=a.blocks)
        pvn.refresh_inference_copy()
        evaluator = pvn.evaluate_leaves_logits  # compact boundary: logits in, the engine gathers the legal priors
    else:
        evaluator = uniform_evaluator
    sp = BatchedSelfPlay(evaluator, B, n_playout=n, seed=0, board_id_base=rank * B, device=local_rank,
                         sampling="device")
    e = sp.engine
    gather = TupleGatherer(512, xdev) if world > 1 else None

    trainer = None
    if a.train_every > 0 and rank == 0:
        from chinesechesszero_amd.replay import ReplayBuffer
        from chinesechesszero_amd.trainer import Trainer
        torch.manual_seed(1)
        trainer = Trainer(PolicyValueNet(device=dev, num_channels=a.channels, resblocks_num=a.blocks))
        rb = ReplayBuffer(32768, dev)
        # no game finishes within a short bench window: prefill the buffer with synthetic rows so that the
        # trainer runs at its steady-state cadence next to self-play ("data": "synthetic")
        g = torch.Generator(device=dev).manual_seed(2)
        ps = torch.rand((8192, 2086), device=dev, generator=g)
        rb.append((torch.rand((8192, 17, 7, 10, 9), device=dev, generator=g) > 0.9).half(), ps / ps.sum(1, keepdim=True),
                  torch.randint(-1, 2, (8192,), device=dev, generator=g).float())
        side = torch.cuda.Stream(device=dev)
        train_steps = [0]

    def per_move():
        sp.finish_move()
        st = e.game_status()
        if st["over"].any() or world > 1:
            s, p, z = e.harvest(max_rows=1 << 21) if st["over"].any() else (e.leaf_input[:0], torch.empty((0, 2086), device=dev), torch.empty((0,), device=dev))
            if gather is not None:
                s, p, z = gather.gather(s.to(xdev), p.to(xdev), z.to(xdev))
            if trainer is not None and s.shape[0]:
                rb.append(s.to(dev), p.to(dev), z.to(dev))

    ev = lambda: torch.cuda.Event(enable_timing=True)
    logits_in = bool(getattr(evaluator, "returns_logits", False))
    import ctypes as _C
    from chinesechesszero_amd._lib import check as check_rc
    ptr_of = lambda t: _C.c_void_p(t.data_ptr())
    step_no = [0]
    state = {"leaf": None}

    def run(steps, timed):
        """steps x [evaluator -> fused k_step (expand+backup of this leaf, select of the next)]; a move boundary
        flushes with expand_backup, plays the move and re-selects."""
        pairs = []
        for _ in range(steps):
            if state["leaf"] is None:
                state["leaf"] = e.select_leaves()
            last_of_move = (step_no[0] + 1) % n == 0
            if timed:
                e0, e1, e2 = ev(), ev(), ev()
                e0.record()
            prob, value = evaluator(state["leaf"])
            if logits_in:  # the softmax+gather of the legal priors belongs to the evaluator side of the split
                e.gather_priors(prob, value)
            if timed:
                e1.record()
            if last_of_move:
                if logits_in:
                    check_rc(e.L.ccz_expand_backup_compact(e.h, e._stream(), ptr_of(value)))
                else:
                    e.expand_backup(prob, value)
                state["leaf"] = None
            else:
                state["leaf"] = e.step_compact(value) if logits_in else e.step(prob, value)
            if timed:
                e2.record()
                if not last_of_move:
                    pairs.append((e0, e1, e2))
            step_no[0] += 1
            if trainer is not None and step_no[0] % a.train_every == 0:
                with torch.cuda.stream(side):
                    trainer.step(*rb.sample(2048), sync=False)
                train_steps[0] += 1
            if last_of_move:
                per_move()
        return pairs

    run(a.warmup, False)
    torch.cuda.synchronize()
    s0 = e.stats()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pairs = run(a.steps, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    s1 = e.stats()
    e.check_healthy()
    # what a HIP-event pair reports around a trivial kernel on this stream (the floor included in avg_launch_us)
    tiny = torch.zeros(64, device=dev)
    fl = []
    for _ in range(64):
        a0, a1 = ev(), ev()
        a0.record()
        tiny.add_(1.0)
        a1.record()
        fl.append((a0, a1))
    torch.cuda.synchronize()
    event_floor_us = float(np.median([x.elapsed_time(y) for x, y in fl])) * 1e3

    # the evaluator's dominant kernel (2 x blocks launches per step): the fused tower convolution, timed live on this stream
    # over the whole tower on the activations of a real leaf batch (ReLU-sparse data clocks higher than dense random data)
    net_roofline = None
    if a.evaluator == "net" and rank == 0 and a.channels == 256 and B >= 192 and os.environ.get("CCZ_FUSED_CONV", "1") != "0":
        import torch.nn.functional as F
        inf = pvn._infer
        leaf = state["leaf"] if state["leaf"] is not None else e.select_leaves()
        with torch.no_grad():
            x = leaf.view(B, 119, 10, 9)
            x = torch.cat([x[:, 49:56], x[:, 105:119]], dim=1).to(torch.float16).contiguous(memory_format=torch.channels_last)
            x0 = inf._epilogue(F.conv2d(x, inf.stem_w, None, padding=1), inf.stem_b)
            inf._tower_fused(x0.clone(memory_format=torch.preserve_format))
            xs = [x0.clone(memory_format=torch.preserve_format) for _ in range(3)]
            c0, c1 = ev(), ev()
            c0.record()
            for xi in xs:
                inf._tower_fused(xi)
            c1.record()
        torch.cuda.synchronize()
        t_conv = c0.elapsed_time(c1) * 1e-3 / (len(xs) * 2 * a.blocks)
        conv_flops = 2.0 * B * 90 * 256 * 256 * 9
        net_roofline = {"bound": "mfma", "kernel": "k_conv3x3_c256 (tower conv3x3 256->256 + bias + residual + ReLU, fp16 in / fp32 acc)",
                        "achieved": conv_flops / t_conv / 1e12, "peak": MFMA_PEAK_F16 / 1e12, "unit": "TFLOP/s",
                        "frac": conv_flops / t_conv / MFMA_PEAK_F16, "traffic": None, "avg_launch_us": t_conv * 1e6,
                        "algorithmic_flops_per_launch": conv_flops, "launches_per_step": 2 * a.blocks,
                        "groups": int(os.environ.get("CCZ_TOWER_GROUPS", "0")) or -(-B // inf.TOWER_GROUP_BOARDS),
                        "note": "a 'launch' is one layer over the whole batch, issued as groups x chains kernel launches over board ranges "
                                "(groups one after the other, the chains of a group concurrently)"}
        per_group = -(-B // net_roofline["groups"])
        net_roofline["chains"] = max(1, min(int(os.environ.get("CCZ_TOWER_CHAINS", inf.TOWER_CHAINS)), 8, per_group // 256))

    sims = s1["sims"] - s0["sims"]
    exp = max(1, s1["expansions"] - s0["expansions"])
    kbar = (s1["sum_children"] - s0["sum_children"]) / exp
    dbar = (s1["sum_depth"] - s0["sum_depth"]) / max(1, sims)
    out = None
    if rank == 0:
        value = world * B * a.steps / dt
        if pairs:
            t_net = float(np.mean([p[0].elapsed_time(p[1]) for p in pairs])) * 1e-3
            t_step = float(np.mean([p[1].elapsed_time(p[2]) for p in pairs])) * 1e-3
        else:
            t_net = t_step = float("nan")
        # algorithmic bytes per simulation, SURVEY 8(d): the fused k_step kernel does all of it. The
        # 21,420-B evaluator input is counted at the 3,780 B that can be non-zero (groups 7/15/16);
        # the 14 static-zero groups are written once at create, not per simulation.
        a_sel = 3780 + 12 * kbar * dbar + 6 * dbar + 180 + 2 * kbar
        a_exp = (4 * kbar + 4) + 18 * kbar + 16 * (dbar + 1)
        a_step = a_sel + a_exp
        a_sim_survey = a_step - 3780 + 21420
        ach = a_step * B / t_step if t_step == t_step else None
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc):
            try:
                with open(pmc) as f:
                    pm = json.load(f)
                traffic = pm.get("k_step", {}).get("hbm_bytes_per_launch")
                if net_roofline is not None:  # HBM bytes of one convolution launch (PMC passes of profiles/run_profile.sh)
                    per_kernel = pm.get("k_conv3x3", {}).get("hbm_bytes_per_launch")
                    net_roofline["traffic"] = per_kernel * net_roofline["chains"] * net_roofline["groups"] if per_kernel else None
            except Exception:
                traffic = None
        flops = 8.551e9 * (a.blocks / 40.0) * (a.channels / 256.0) ** 2
        out = {
            "metric": "self-play MCTS simulations/sec", "value": value, "unit": "sims/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 rules / f32 Q / f64 PUCT (net: fp16)", "data": "synthetic",
            "config": {"workload": f"{B} concurrent boards/GPU x {n} sims/move, Dirichlet root noise on (device Philox), "
                                   f"{'random-init %dx%d policy-value net fp16' % (a.blocks, a.channels) if a.evaluator == 'net' else 'stub evaluator (uniform priors, v=0)'}"
                                   ", all boards from the opening position",
                       "boards_per_gpu": B, "sims_per_move": n, "evaluator": a.evaluator},
            "moves_per_sec": value / n,
            "roofline": {"bound": "hbm", "kernel": "k_step (fused expand+backup+select+movegen+encode)",
                         "achieved": (ach or 0) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": (ach or 0) / HBM_PEAK,
                         "traffic": traffic, "algorithmic_bytes_per_launch": a_step * B, "avg_launch_us": t_step * 1e6,
                         "k_bar": kbar, "d_bar": dbar, "event_floor_us": event_floor_us},
            "survey_a_sim_bytes": a_sim_survey,
            "step_split_us": {"k_step": t_step * 1e6, "evaluator": t_net * 1e6},
            "trainer_updates": (train_steps[0] if trainer is not None else 0),
            "net_roofline": net_roofline,
            "net_tflops": (flops * B / t_net / 1e12) if (a.evaluator == "net" and t_net == t_net) else None,
            "engine_hbm_gb": s1["hbm_bytes"] / 1e9, "nodes_peak": s1["nodes_peak"], "depth_peak": s1["depth_peak"],
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_baseline_seconds, a.blocks, a.channels)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
