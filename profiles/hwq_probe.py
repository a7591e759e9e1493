#!/usr/bin/env python3
"""Hardware-queue aliasing probe (round 5). The 40x256 evaluator on 4096 synthetic rows, timed in a process that also holds the streams of the
multi-GPU exchange. argv: group (0|1) order (exchange_first|eval_first|init_only) ; env GPU_MAX_HW_QUEUES as given (the package default is 8).
  exchange_first: process group + a USED exchange (side stream, RCCL stream) before the evaluator's first evaluation
  eval_first:     process group initialised, evaluator used, THEN the exchange used (bench.py's order: the first boundary comes after the first step)
  init_only:      process group initialised, exchange never used
  net_then_exchange: process group, THEN the net is built (refresh_inference_copy binds the launch-chain streams: net.chain_streams), THEN the
                  exchange is used, THEN the first evaluation -- the order of the collector CLI (weights broadcast before the first step)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("CCZ_NO_QUEUE_DEFAULT"):      # measure HIP's own default (4): keep the package from setting 8
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
import chinesechesszero_amd  # noqa: E402,F401
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    group, order = sys.argv[1] == "1", (sys.argv[2] if len(sys.argv) > 2 else "exchange_first")
    from chinesechesszero_amd.launch import free_port
    from chinesechesszero_amd.net import PolicyValueNet
    from chinesechesszero_amd.replay import AsyncRecordExchange
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ex = None

    def use_exchange():
        nonlocal ex
        ex = ex or AsyncRecordExchange(4096, dev, always_collective=True, timeout_s=60)
        ex.post([torch.zeros((64, 880), dtype=torch.uint8, device=dev)], games=1)
        for _ in ex.flush_iter():
            pass

    if group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        if order == "exchange_first":
            use_exchange()
    torch.manual_seed(0)
    pvn = PolicyValueNet(device=dev, num_channels=256, resblocks_num=40)
    pvn.refresh_inference_copy()
    if group and order == "net_then_exchange":
        use_exchange()
    g = torch.Generator(device=dev).manual_seed(3)
    leaf = (torch.rand((4096, 17, 7, 10, 9), device=dev, generator=g) > 0.9).half()

    def t_eval(n=16):
        for _ in range(4):
            pvn.evaluate_leaves_logits(leaf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            pvn.evaluate_leaves_logits(leaf)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    out = {"group": group, "order": order if group else None, "queues": os.environ.get("GPU_MAX_HW_QUEUES"), "ms": [t_eval()]}
    if group and order == "eval_first":
        use_exchange()
    out["ms"].append(t_eval())
    out["ms"].append(t_eval())
    print("HWQ_PROBE", json.dumps(out))
    if group:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
