"""Loader of tests/golden/rules_kat.json (hand-derived known answers for the rule statements of DESIGN.md section 4)."""
import json
import os

import numpy as np

_PC = {"p": 1, "c": 2, "r": 3, "n": 4, "b": 5, "a": 6, "k": 7}
PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rules_kat.json")


def sq(name: str) -> int:
    return (ord(name[0]) - 97) + 9 * int(name[1])


def cases():
    with open(PATH, encoding="utf-8") as f:
        return json.load(f)["cases"]


def start_of(case):
    """(squares uint8[90], turn 1 RED / 0 BLACK, halfmove clock)"""
    b = np.zeros(90, dtype=np.uint8)
    for name, ch in case["pieces"].items():
        b[sq(name)] = _PC[ch.lower()] + (0 if ch.isupper() else 8)
    return b, 1 if case["turn"] == "red" else 0, int(case.get("halfmove", 0))


def checks_of(case):
    """[(number of moves played, expectations)], ascending"""
    n = len(case.get("moves", []))
    return sorted(((int(c.get("after", n)), c) for c in case["checks"]), key=lambda t: t[0])


WINNER = {"red": True, "black": False, None: None}
