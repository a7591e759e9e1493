"""CPU: what stands between `bench.py --gpus N` and a hang on its first real multi-GPU node -- the GPU-free pre-flight, the
rank guard that turns any exception into a dead process (so that the launcher tears the job down) instead of peers waiting in
a collective, the exchange's abort flag, and the one-buffer weight broadcast. gloo, world size 2."""
import os
import socket
import subprocess
import sys
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _topology(tmp_path, simd_counts):
    for i, n in enumerate(simd_counts):
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {0 if n else 64}\nsimd_count {n}\nmem_banks_count 1\n")
    return str(tmp_path)


def test_preflight_counts_gpus_without_touching_hip(tmp_path, monkeypatch):
    from chinesechesszero_amd import launch
    for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    root = _topology(tmp_path, [0, 0, 1024, 1024, 1024, 1024])          # two CPU nodes, four GPUs
    assert launch.visible_gpus(root) == 4
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1")
    assert launch.visible_gpus(root) == 2                               # a device mask clips the count
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    assert launch.preflight(4, gpus=4) is None and launch.preflight(1, gpus=1) is None
    why = launch.preflight(8, gpus=4)
    assert why and "\n" not in why and "--gpus 8" in why and "shows 4" in why      # ONE line that says what is missing
    assert launch.preflight(2, share_gpu=True, gpus=1) is None                     # the gloo rehearsal: all ranks on one GPU
    assert launch.preflight(2, share_gpu=True, gpus=0) is not None


def test_self_launch_refuses_before_starting_any_rank():
    """`python bench.py --gpus 8` on a node without that many GPUs: non-zero exit and one line, within seconds, no child."""
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], cwd=ROOT, capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        pytest.skip("this node really has 8 GPUs")
    assert r.returncode != 0 and "not starting any rank" in r.stderr and time.time() - t0 < 120
    assert '"metric"' not in r.stdout


_FAULT = r"""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from chinesechesszero_amd import launch
def main():
    launch.init_distributed("gloo", timeout_s=300)          # the job must NOT need this timeout to end
    rank = dist.get_rank()
    x = torch.zeros(4)
    dist.all_reduce(x)                                       # the window opens: everybody is here
    if rank == 1:
        raise RuntimeError("engine error flags 1: node pool exhausted")   # what check_healthy() raises mid-window
    buf = torch.zeros(2 * 8, dtype=torch.uint8)
    dist.all_gather_into_tensor(buf, torch.zeros(8, dtype=torch.uint8))   # rank 0 waits here for a peer that is gone
    print("RANK0_PASSED_THE_COLLECTIVE")
    return 0
sys.exit(launch.guarded(main))
"""


def test_a_rank_that_raises_inside_the_window_ends_the_job_within_seconds(tmp_path):
    """Rank 1 raises while rank 0 sits in the all-gather. `guarded` prints the rank and ends the process with os._exit(1); the
    launcher sees a dead rank and tears the group down: the parent exits non-zero in well under the collective's timeout."""
    script = tmp_path / "fault.py"
    script.write_text(_FAULT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script), ROOT]
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280)
    took = time.time() - t0
    assert r.returncode != 0, (r.stdout[-1000:], r.stderr[-2000:])
    assert took < 60, took
    assert "[rank 1] failed" in r.stderr and "node pool exhausted" in r.stderr
    assert "RANK0_PASSED_THE_COLLECTIVE" not in r.stdout


def _abort_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from test_cpu_distributed import _RecordSource, _records
        from chinesechesszero_amd.replay import GatherAborted, RecordGatherer, exchange_finished_games
        g = RecordGatherer(8, "cpu")
        # rank 1 holds ONE game of 11 plies: longer than the 8-ply slot. Rank 0 holds an ordinary game.
        mine = _records(rank, (11,) if rank == 1 else (5,))
        try:
            g.gather(mine)
            q.put((rank, "no error"))
        except GatherAborted as e:
            q.put((rank, "aborted", str(e)))
        # the same misconfiguration caught BEFORE any collective when the source says how long its games can get
        src = _RecordSource(rank, ((3,), (3,))[rank])
        src.max_plies = 64
        try:
            list(exchange_finished_games(src, g, 1))
            q.put((rank, "no error"))
        except ValueError as e:
            q.put((rank, "refused", str(e)))
        dist.barrier()     # both ranks are still in step: nobody was left behind in a collective
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, "crash", traceback.format_exc()[-1500:]))


def test_a_game_longer_than_the_slot_makes_every_rank_raise_together():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_abort_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2 * world)]
    for p in procs:
        p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    aborted = sorted(r for r in res if r[1] == "aborted")
    refused = sorted(r for r in res if r[1] == "refused")
    assert [r[0] for r in aborted] == [0, 1], res                      # BOTH ranks raised, after the same collective
    assert "longer than the exchange slot" in aborted[1][2] and "rank(s) [1]" in aborted[0][2]
    assert [r[0] for r in refused] == [0, 1] and "max_plies" in refused[0][2]


def _bcast_worker(rank, world, port, what, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from chinesechesszero_amd import replay
        from chinesechesszero_amd.net import PolicyValueNet
        torch.manual_seed(10 + rank)  # different weights on every rank before the reload
        pvn = PolicyValueNet(use_gpu=False, device="cpu", num_channels=256, resblocks_num=1)
        pvn.refresh_inference_copy()
        addr = [p.data_ptr() for p in pvn._infer.parameters()]
        v0 = pvn.weights_version
        calls = []
        orig = dist.broadcast
        dist.broadcast = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            replay.broadcast_model(pvn, src=0, what=what)
        finally:
            dist.broadcast = orig
        inf = pvn._infer
        dig = float(sum(p.double().abs().sum() for p in inf.parameters()))
        g16 = float(sum(p.double().abs().sum() for n, p in inf.named_parameters() if "g16" in n))
        same_addr = addr == [p.data_ptr() for p in inf.parameters()]
        # the packed copies are exactly what InferenceNet says it derives (one list for the sender's skip and the receiver's rebuild)
        assert {n for n, _ in inf.named_parameters() if "g16" in n} == inf.derived_parameter_names()
        # after an inference-only reload a RECEIVER's fp32 net is stale: rebuilding the copy from it or saving it is refused
        guarded = None
        if what == "inference":
            guarded = []
            from chinesechesszero_amd.trainer import Trainer
            batch = (torch.zeros(2, 119, 10, 9), torch.full((2, 2086), 1.0 / 2086), torch.zeros(2))
            for call in (pvn.refresh_inference_copy, lambda: pvn.save_model(os.devnull), lambda: pvn.train_step(*batch),
                         lambda: Trainer(pvn).step(*batch)):      # ... and training the old weights
                try:
                    call()
                    guarded.append(False)
                except RuntimeError as e:
                    guarded.append("broadcast_model(what='state')" in str(e))
            replay.broadcast_model(pvn, src=0, what="state")     # ... until the state itself is sent
            pvn.save_model(os.devnull)
        q.put((rank, True, len(calls), dig, g16, same_addr, pvn.weights_version > v0, guarded))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, False, traceback.format_exc()[-1500:]))


@pytest.mark.parametrize("what", ["state", "inference"])
def test_weight_broadcast_is_one_collective_of_one_flat_buffer(what):
    """Round 3 issued one broadcast per tensor (~500 at 40 blocks). Now: ONE byte buffer whatever the dtypes. "state" sends the
    fp32 net (every rank can save / train), "inference" only the BN-folded fp16 copy, written in place (addresses unchanged) with
    the packed group-of-16 weights re-derived locally."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bcast_worker, args=(r, world, port, what, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] for r in res), res
    assert res[0][2] == res[1][2] == 1                         # one collective
    assert res[0][3] == res[1][3] and res[0][4] == res[1][4] and res[0][4] > 0   # same inference weights, packed copies included
    assert all(r[6] for r in res)                              # weights_version moved: evaluation caches keyed to it are emptied
    if what == "inference":
        assert res[1][5]                                       # in place: what captured hipGraphs point at is still valid
        assert res[1][7] == [True] * 4 and res[0][7] == [False] * 4   # the receiver is guarded, the source (whose fp32 IS the truth) is not
