#!/usr/bin/env python3
"""Does an initialised RCCL process group slow the evaluator down? (round 5: `bench.py --rccl-group-of-one` ran the full-size window at
26.6 ms/step against 21.2 plain, the tower convolution 330 us per layer against 262.) One process, one GPU: the 40x256 evaluator on
4096 synthetic rows, timed (a) before any process group exists, (b) after init_process_group on the backend under test, (c) after one
collective, (d) after destroy_process_group. argv: backend (nccl|gloo), device_id (1 = pass device_id= to init: eager communicator)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
    eager = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
    import torch.distributed as dist
    from chinesechesszero_amd.launch import free_port
    from chinesechesszero_amd.net import PolicyValueNet
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    pvn = PolicyValueNet(device=dev, num_channels=256, resblocks_num=40)
    pvn.refresh_inference_copy()
    g = torch.Generator(device=dev).manual_seed(3)
    leaf = (torch.rand((4096, 17, 7, 10, 9), device=dev, generator=g) > 0.9).half()

    def t_eval(n=24):
        for _ in range(4):
            pvn.evaluate_leaves_logits(leaf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            pvn.evaluate_leaves_logits(leaf)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    out = {"backend": backend, "device_id_passed": eager, "env": {k: v for k, v in os.environ.items() if k.startswith(("NCCL", "RCCL", "TORCH_NCCL", "HSA_", "GPU_MAX", "HIP_"))}}
    out["ms_before_any_process_group"] = t_eval()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(free_port()))
    kw = {"device_id": dev} if (eager and backend == "nccl") else {}
    dist.init_process_group(backend, rank=0, world_size=1, **kw)
    out["ms_after_init_process_group"] = t_eval()
    x = torch.ones(1 << 20, device=dev if backend == "nccl" else "cpu")
    y = torch.empty_like(x)
    dist.all_gather_into_tensor(y, x)
    torch.cuda.synchronize()
    out["ms_after_one_collective"] = t_eval()
    w = dist.all_gather_into_tensor(y, x, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    out["ms_after_an_async_collective"] = t_eval()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        w = dist.all_gather_into_tensor(y, x, async_op=True)
        w.wait()
    torch.cuda.synchronize()
    out["ms_after_a_collective_from_a_side_stream"] = t_eval()
    dist.destroy_process_group()
    out["ms_after_destroy_process_group"] = t_eval()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
