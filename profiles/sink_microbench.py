#!/usr/bin/env python3
"""What one move of 4096 boards costs the collector's HOST at the sink (round 5): ~14.2 k finished plies = 28.4 k dense rows.
dense: TupleSink.append of the rows (float16 planes, float32 -> float64 pi, winners) -- what rounds 1-4 wrote while collecting;
records: TupleSink.append_records of the same games as compact ply records (880 B per ply); and the finalize() that expands them."""
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from chinesechesszero_amd.collect import TupleSink
    plies = 14200
    out = {"plies_per_move": plies, "rows_per_move": 2 * plies}
    d = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        s = TupleSink(os.path.join(d, "dense"))
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        states = torch.zeros((2 * plies, 17, 7, 10, 9), dtype=torch.float16, device=dev)
        pi = torch.rand((2 * plies, 2086), device=dev)
        z = torch.zeros(2 * plies, device=dev)
        t0 = time.perf_counter()
        s.append(states, pi, z, games=100)
        out["dense_append_s"] = time.perf_counter() - t0
        out["dense_shard_mb"] = sum(os.path.getsize(os.path.join(d, "dense", f)) for f in os.listdir(os.path.join(d, "dense"))) / 1e6
        s.close()
        del states, pi, z
        if dev == "cuda":   # real records of real games, so that finalize() can expand them
            from chinesechesszero_amd.net import uniform_evaluator
            from chinesechesszero_amd.selfplay import BatchedSelfPlay
            sp = BatchedSelfPlay(uniform_evaluator, 1024, n_playout=2, seed=1, max_plies=14)
            for _ in range(15):
                sp.run_move()
            rec = torch.cat(list(sp.harvest_record_chunks(1 << 16)))
            flags, pot = sp.engine.record_flags(), sp.engine.plane_of_type
            rec = rec[:plies] if rec.shape[0] >= plies else rec
        else:
            rec, flags, pot = torch.zeros((plies, 880), dtype=torch.uint8), 0, None
        out["records"] = int(rec.shape[0])
        s = TupleSink(os.path.join(d, "records"))
        t0 = time.perf_counter()
        s.append_records(rec, flags, pot, games=100)
        out["records_append_s"] = time.perf_counter() - t0
        out["records_shard_mb"] = sum(os.path.getsize(os.path.join(d, "records", f)) for f in os.listdir(os.path.join(d, "records"))) / 1e6
        if dev == "cuda":
            t0 = time.perf_counter()
            n = s.finalize()
            out["records_finalize_s"] = time.perf_counter() - t0
            out["rows_after_finalize"] = n
        s.close()
    finally:
        shutil.rmtree(d, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
