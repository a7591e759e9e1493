"""perft(6) of the Xiangqi start position, twice: by the HIP kernels alone (tests/test_gpu_rules.py gpu_perft: ccz_legal_moves +
ccz_apply_moves over the 133 M depth-5 positions) and by the independently written CPU oracle (oracle/xq_rules.c xq_perft, the
1,920 depth-2 positions spread over the host cores), compared first move by first move ("divide": 44 subtotals).

usage (GPU box): python profiles/perft6.py [--workers 16] > gpurun_out/r06_perft6.json
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _oracle_task(args):
    first, second = args
    from oracle import OracleBoard
    b = OracleBoard()
    b.push_id(first)
    b.push_id(second)
    return first, b.perft(4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=min(16, os.cpu_count() or 1))
    ap.add_argument("--depth", type=int, default=6)
    a = ap.parse_args()
    assert a.depth == 6, "the oracle leg is written for depth 6 (perft(4) below every depth-2 position)"
    from oracle import OracleBoard
    root = OracleBoard()
    firsts = root.legal_ids()
    tasks = []
    for f in firsts:
        c = root.copy()
        c.push_id(f)
        tasks += [(f, s) for s in c.legal_ids()]
    t0 = time.time()
    sub = {f: 0 for f in firsts}
    with ProcessPoolExecutor(max_workers=a.workers) as ex:   # started before anything touches the GPU in this process
        for f, n in ex.map(_oracle_task, tasks, chunksize=8):
            sub[f] += n
    t_oracle = time.time() - t0
    from test_gpu_rules import gpu_perft
    t0 = time.time()
    counts, divide = gpu_perft(a.depth)
    t_gpu = time.time() - t0
    div_oracle = [sub[f] for f in sorted(firsts)]
    out = {"what": "perft(6) of the start position: HIP kernels (ccz_legal_moves / ccz_apply_moves) against the CPU oracle (xq_perft), split by the first move (ascending move id)",
           "perft_gpu": counts, "perft6_oracle": sum(div_oracle), "first_moves": sorted(firsts),
           "divide_gpu": divide, "divide_oracle": div_oracle, "equal": divide == div_oracle,
           "commonly_cited_value": 5392831844, "seconds_gpu": round(t_gpu, 1), "seconds_oracle": round(t_oracle, 1), "oracle_workers": a.workers}
    print(json.dumps(out))
    return 0 if out["equal"] and counts[-1] == 5392831844 else 1


if __name__ == "__main__":
    sys.exit(main())
