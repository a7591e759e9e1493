"""How the positions of the double-sided perpetual-check cases of rules_kat.json were FOUND (they are then verified on paper; the file
holds the hand-derived expectations, not this script's output): random small-material placements, red to move, searched with the
oracle for a 4-ply cycle back to the start position in which

    mutual   every move gives check (the side to move is in check at every ply), or
    partial  both red moves and the first black move give check, the second black move does not.

    python tests/golden/find_check_cycles.py mutual 3000000     # ~1 minute on 8 cores: ~20 cycles
"""
import json
import os
import random
import sys
import time
from multiprocessing import Pool

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import OracleBoard, move_table  # noqa: E402

PC = {"p": 1, "c": 2, "r": 3, "n": 4, "b": 5, "a": 6, "k": 7}
RED_PALACE = [f + 9 * r for r in range(3) for f in range(3, 6)]
BLACK_PALACE = [f + 9 * r for r in range(7, 10) for f in range(3, 6)]
MATERIAL = [("RC", "RC"), ("RN", "RN"), ("RC", "RN"), ("RR", "RR"), ("RCN", "RCN"), ("RC", "R"), ("RRC", "RRC"), ("CC", "CC"), ("RCC", "RCC")]


def cycle(sq, mutual):
    b0 = OracleBoard.from_array(sq, 1, 0)
    if b0.in_check() != mutual:
        return None
    start = sq.tobytes()

    def rec(b, ply, line):
        for mid in b.legal_ids():
            c = b.copy()
            c.push_id(mid)
            if c.in_check() != (mutual or ply != 3):
                continue
            if ply == 3:
                if c.squares().tobytes() == start:
                    return line + [mid]
                continue
            r = rec(c, ply + 1, line + [mid])
            if r:
                return r
        return None
    return rec(b0, 0, [])


def worker(args):
    seed, n, mutual = args
    rnd = random.Random(seed)
    found = []
    for _ in range(n):
        red, black = rnd.choice(MATERIAL)
        sq = np.zeros(90, np.uint8)
        sq[rnd.choice(RED_PALACE)] = 7
        sq[rnd.choice(BLACK_PALACE)] = 15
        for ch in red + black.lower():
            s = rnd.randrange(90)
            while sq[s]:
                s = rnd.randrange(90)
            sq[s] = PC[ch.lower()] + (0 if ch.isupper() else 8)
        try:
            line = cycle(sq, mutual)
        except Exception:      # a placement the oracle refuses (kings facing, ...)
            continue
        if line:
            found.append((sq.tolist(), line))
            if len(found) >= 3:
                break
    return found


if __name__ == "__main__":
    mutual = (sys.argv[1] if len(sys.argv) > 1 else "mutual") == "mutual"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
    t0 = time.time()
    with Pool(7) as pool:
        res = pool.map(worker, [(s, n, mutual) for s in range(7)])
    found = [f for r in res for f in r]
    print(f"{len(found)} cycles in {time.time() - t0:.0f} s", file=sys.stderr)
    names = move_table()
    for sq, line in found:
        pieces = {"abcdefghi"[i % 9] + str(i // 9): ("?PCRNBAK"[v & 7] if v < 8 else "?pcrnbak"[v & 7]) for i, v in enumerate(sq) if v}
        print(json.dumps({"pieces": pieces, "turn": "red", "moves": [names[m] for m in line]}))
