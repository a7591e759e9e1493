"""The one-game evaluator as the GPU sees it: durations of its kernels and the gaps between them, from a kernel trace
(rocprofv3 --kernel-trace --output-format csv -- python3 profiles/single_board_scouts.py).

    python profiles/small_timeline.py <kernel_trace.csv>
"""
import collections
import csv
import json
import sys

import numpy as np


def main(path):
    rows = []
    with open(path) as f:
        for x in csv.DictReader(f):
            rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"].split("(")[0][-48:],
                         int(x["Grid_Size_X"]) * int(x["Grid_Size_Y"]) // max(1, int(x["Workgroup_Size_X"]))))
    rows.sort()
    rows = rows[len(rows) // 2:]                      # the timed moves are the later half of the run
    dur = collections.defaultdict(list)
    gap = collections.defaultdict(list)
    for a, b in zip(rows, rows[1:]):
        dur[a[2]].append((a[1] - a[0]) / 1e3)
        g = (b[0] - a[1]) / 1e3
        if g < 30:
            gap[a[2] + " -> " + b[2]].append(g)
    out = {"kernels": {k: {"n": len(v), "workgroups": None, "mean_us": round(float(np.mean(v)), 2), "median_us": round(float(np.median(v)), 2)}
                       for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:14]},
           "gaps_us": {k: {"n": len(v), "mean": round(float(np.mean(v)), 2), "median": round(float(np.median(v)), 2)}
                       for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1]))[:10]}}
    for k in out["kernels"]:
        out["kernels"][k]["workgroups"] = collections.Counter(r[3] for r in rows if r[2] == k).most_common(1)[0][0]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
