"""Does replaying the lockstep step as ONE hipGraph buy anything at 1024 boards (BASELINE configs[1])? The step -- cache probe + plan,
the evaluator on the planned rows (three launch chains on their own streams), softmax + gather + store, the fused simulator kernel -- is
launched eagerly by the product (the host is ~4x ahead of the GPU at this size); here the same body is captured once and replayed.

    python profiles/graph_step_probe.py [boards] > profiles/r06_graph_step_1024.json
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(B=1024, K=120):
    from chinesechesszero_amd.net import PolicyValueNet
    from chinesechesszero_amd.selfplay import BatchedSelfPlay
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    pvn = PolicyValueNet(device=dev)
    pvn.refresh_inference_copy()
    ev = pvn.evaluate_leaves_logits
    sp = BatchedSelfPlay(ev, B, n_playout=400, seed=1, eval_cache_log2=24, max_plies=200)
    e = sp.engine
    sp.run_move()                                    # one whole move: trees, cache and allocator in their steady state
    e.select_leaves()                                # the next move's first leaf is pending

    def body():
        plan = e.eval_plan()
        logits, value = ev(e.leaf_input, plan=plan)
        e.step_planned(logits, value)

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    out = {"what": __doc__.split("\n\n")[0], "boards": B, "steps_per_leg": K}
    for _ in range(8):
        body()
    out["eager_ms_per_step_1"] = timed(body, K)
    try:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        out["graph_ms_per_step"] = timed(g.replay, K)
        out["capture"] = "ok"
    except Exception as ex:  # noqa: BLE001 -- the probe reports what the capture said
        out["capture"] = f"failed: {type(ex).__name__}: {str(ex)[:300]}"
    out["eager_ms_per_step_2"] = timed(body, min(K, 20))
    st = e.stats()
    out["error_flags"] = st["error_flags"]
    out["sims_per_s"] = {k: B / (out[k] * 1e-3) for k in ("eager_ms_per_step_1", "graph_ms_per_step", "eager_ms_per_step_2") if k in out}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1024)
