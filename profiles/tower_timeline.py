"""How full the tower keeps the chip: the kernel trace of one bench run (rocprofv3 --kernel-trace --output-format csv) reduced to
the launch structure of the group-of-16 tower kernels -- kernels in flight over time, gaps between the dependent launches of a chain,
kernel durations -- and, from the live tiles the bench line reports, the CU time one tile takes when the chip is counted as full.

    python profiles/tower_timeline.py <kernel_trace.csv> <bench.json> [window_ms]

Run by profiles/calls/r06_call14.sh (1024 and 4096 boards on one box); output: profiles/r06_tower_timeline.json.
"""
import collections
import csv
import json
import sys

import numpy as np


def main(trace, bench, window_ms=120.0):
    line = json.loads([l for l in open(bench) if l.startswith("{")][-1])
    rows = []
    with open(trace) as f:
        for x in csv.DictReader(f):
            name = x["Kernel_Name"]
            if "k_conv3x3_g16" not in name:
                continue
            kind = "edge" if "g16_edge" in name else "heads" if "g16_heads" in name else "stem" if "g16_stem" in name else "tower"
            rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), kind, int(x["Queue_Id"]),
                         int(x["Grid_Size_X"]) // max(1, int(x["Workgroup_Size_X"]))))
    rows.sort()
    t_end = rows[-1][1]
    w = [x for x in rows if t_end - (window_ms + 10.0) * 1e6 < x[0] < t_end - 10.0e6]  # the timed steps are the last ones of the run
    tower = [x for x in w if x[2] in ("tower", "edge")]
    out = {"boards": line["config"].get("boards_per_gpu"), "sims_per_s": line["value"], "ms_per_step": line["ms_per_step"],
           "net_roofline_frac": (line.get("net_roofline") or {}).get("frac"),
           "rows_computed_per_step": ((line.get("eval_cache") or {}).get("rows_computed_per_step")),
           "window_ms": window_ms, "tower_kernels_in_window": len(tower),
           "grids": dict(collections.Counter(f"{x[2]}:{x[4]}" for x in tower).most_common(6)), "per_queue": {}}
    for q in sorted({x[3] for x in tower}):
        cq = [x for x in tower if x[3] == q]
        gaps = np.array([b[0] - a[1] for a, b in zip(cq, cq[1:]) if b[0] - a[1] < 20e3]) / 1e3
        durs = np.array([x[1] - x[0] for x in cq]) / 1e3
        out["per_queue"][str(q)] = {"launches": len(cq), "duration_us": {"mean": round(float(durs.mean()), 2), "median": round(float(np.median(durs)), 2),
                                                                       "p10": round(float(np.percentile(durs, 10)), 2), "p90": round(float(np.percentile(durs, 90)), 2)},
                                    "gap_to_next_launch_of_the_chain_us": {"mean": round(float(gaps.mean()), 3), "median": round(float(np.median(gaps)), 3),
                                                                          "p99": round(float(np.percentile(gaps, 99)), 3)}}
    ev = []
    for s, e, _, _, g in tower:
        ev.append((s, 1, g))
        ev.append((e, -1, -g))
    ev.sort()
    n = grid = 0
    last = ev[0][0]
    in_flight = collections.Counter()
    wg = collections.Counter()
    for t, dn, dg in ev:
        in_flight[n] += t - last
        wg["0" if grid == 0 else "1-127" if grid < 128 else "128-255" if grid < 256 else ">=256"] += t - last
        last = t
        n += dn
        grid += dg
    tot = sum(in_flight.values())
    out["share_of_time_by_tower_kernels_in_flight"] = {str(k): round(v / tot, 4) for k, v in sorted(in_flight.items())}
    out["share_of_time_by_launched_workgroups_in_flight"] = {k: round(v / tot, 4) for k, v in wg.items()}
    # CU time per live tile with the chip counted as full while >= 256 workgroups are in flight: 256 CUs x that time / live tiles computed
    rows_step = out["rows_computed_per_step"]
    if rows_step:
        busy = sum(v for k, v in wg.items() if k == ">=256") / 1e3  # us
        groups = -(-int(round(rows_step)) // 16)
        layers = 80
        live_tiles = groups * 5 * layers * (tot / 1e6 / line["ms_per_step"])
        out["cu_us_per_live_tile_chip_counted_full"] = round(256 * busy / live_tiles, 2)
        out["what"] = ("cu_us_per_live_tile = 256 CUs x time with >= 256 tower workgroups launched / (5 tiles x ceil(rows / 16) groups x 80 layers x steps in the window); "
                       "edge-pair launches (4096 boards) count as tiles of the same group, so the figure there is per 5-tile group / 5")
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 120.0)
