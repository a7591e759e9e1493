"""One process per GPU: what a multi-rank run of this package needs around ``torch.distributed`` so that its FIRST contact with a
real multi-GPU node fails fast and loud instead of hanging (bench.py, collect.py's multi-rank collector).

Nothing in this module touches HIP: a parent that spawns the ranks must stay GPU-free (replacing or forking a process that has
initialised the GPU is refused on the GPU boxes). The reference has no counterpart: its N collectors are N independent shell
commands appending to one file (collect.py:181-183, README.md:31-48).
"""
from __future__ import annotations

import glob
import os
import socket
import subprocess
import sys
import traceback
from datetime import timedelta

DIST_TIMEOUT_S = 180  # a rank that dies leaves its peers in a collective: they give up after this, not after torch's 10 minutes


def _mask_len(name: str):
    v = os.environ.get(name)
    if v is None:
        return None
    return len([x for x in v.split(",") if x.strip() != ""])


def visible_gpus(sysfs_root: str = "/sys/class/kfd/kfd/topology/nodes") -> int:
    """GPUs a child process of this one could open, WITHOUT initialising HIP here: KFD topology nodes with SIMDs (CPUs are
    nodes with ``simd_count 0``), clipped by the ``HIP_VISIBLE_DEVICES`` / ``ROCR_VISIBLE_DEVICES`` / ``CUDA_VISIBLE_DEVICES``
    masks. When the topology cannot be read (no /sys in a sandbox), a short-lived CHILD process asks
    ``torch.cuda.device_count()`` (:func:`_device_count_in_child`): this process never calls into HIP."""
    n = None
    try:
        nodes = glob.glob(os.path.join(sysfs_root, "*", "properties"))
        if nodes:
            n = 0
            for p in nodes:
                try:
                    with open(p) as f:
                        for line in f:
                            if line.startswith("simd_count"):
                                n += 1 if int(line.split()[1]) > 0 else 0
                                break
                except OSError:
                    pass   # a node this user may not read is not a GPU this user may open
    except Exception:
        n = None
    if n is None:
        n = _device_count_in_child()
    for name in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        m = _mask_len(name)
        if m is not None:
            n = min(n, m)
    return n


def _device_count_in_child(timeout_s: float = 120.0) -> int:
    """``torch.cuda.device_count()`` as a fresh child process sees it. On ROCm that call can go through ``hipGetDeviceCount`` and
    initialise HIP/HSA; the parent that later spawns the ranks must stay GPU-free (module docstring), so the question is asked in
    a process that exits right after answering. 0 when the child fails or says nothing parseable."""
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print('ccz_device_count', torch.cuda.device_count())"],
                           capture_output=True, text=True, timeout=timeout_s)
        for line in r.stdout.splitlines():
            if line.startswith("ccz_device_count"):
                return int(line.split()[1])
    except Exception:
        pass
    return 0


def board_partition(world: int, boards: int, boards_rank0: int | None = None):
    """Boards per rank and the global id of every rank's first board. All ranks hold ``boards`` boards except rank 0, which may
    hold fewer (``boards_rank0``: the rank that shares its GPU with the trainer, BASELINE configs[4]). The ids are a prefix sum, so
    board ``g`` of the job has the same RNG stream -- and plays the same games -- however the boards are split over ranks."""
    counts = [int(boards)] * int(world)
    if boards_rank0 is not None:
        counts[0] = int(boards_rank0)
    if min(counts) <= 0:
        raise ValueError("every rank needs at least one board")
    bases, acc = [], 0
    for c in counts:
        bases.append(acc)
        acc += c
    return counts, bases


def preflight(n_ranks: int, share_gpu: bool = False, gpus: int | None = None) -> str | None:
    """None if ``n_ranks`` ranks can each have a GPU (``share_gpu``: one GPU for all, the gloo rehearsal), else ONE line saying
    why not -- the parent prints it and exits non-zero before any child exists."""
    have = visible_gpus() if gpus is None else int(gpus)
    need = 1 if share_gpu else int(n_ranks)
    if have < need and gpus is None:
        # second opinion before refusing: a container may show less of the KFD topology than its processes can open. Asked in a
        # short-lived child: counting devices can initialise HIP, and this process must not
        have = max(have, _device_count_in_child())
    if have < need:
        return (f"bench: --gpus {n_ranks} needs {need} visible GPU(s), this node shows {have} "
                f"(KFD topology / *_VISIBLE_DEVICES); not starting any rank")
    return None


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(script: str, n: int, argv, share_gpu: bool = False, cores: int | None = None) -> int:
    """``python script --gpus N`` without a launcher: run ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N script
    <argv>`` as a CHILD process (one rank per GPU over RCCL) and return its exit code; rank 0's output goes to the inherited
    stdout. Called before anything has initialised the GPU in this process, which only waits. Never an exec: replacing a process
    is refused on the GPU boxes, and a child keeps the exit code honest. A node with too few GPUs is refused here, in one line."""
    why = preflight(n, share_gpu)
    if why:
        print(why, file=sys.stderr, flush=True)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(script)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's intra-node transport needs it on these hosts
    env.setdefault("OMP_NUM_THREADS", str(max(1, (cores or os.cpu_count() or 1) // n)))
    return subprocess.call(cmd, env=env)


def init_distributed(backend: str, device=None, timeout_s: int = DIST_TIMEOUT_S):
    """``init_process_group`` with a timeout sized for a benchmark, not for a training job: collectives (gloo) and the
    watchdog (nccl = RCCL) give up after ``timeout_s`` seconds."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = {"timeout": timedelta(seconds=int(timeout_s))}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, **kw)
    return dist


def guarded(main):
    """Run one rank's ``main()``. Any exception -- a HIP error, an engine error flag, an assertion -- prints the rank and the
    traceback and ends the PROCESS with ``os._exit(1)``: no interpreter shutdown that waits for the process group's destructor
    while the peers sit in a collective. ``torch.distributed.run`` sees the dead rank and tears the others down; the job's exit
    code is non-zero within seconds instead of after the collective's timeout."""
    try:
        rc = main()
    except SystemExit as e:
        code = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
        if code == 0:
            return 0
        if e.code is not None and not isinstance(e.code, int):
            print(e.code, file=sys.stderr)
        _die(code)
    except BaseException:
        rank = os.environ.get("RANK", "0")
        print(f"[rank {rank}] failed:\n{traceback.format_exc()}", file=sys.stderr, flush=True)
        _die(1)
    return rc or 0


def _die(code: int):
    try:
        sys.stdout.flush()
        sys.stderr.flush()
    finally:
        os._exit(code if code else 1)
