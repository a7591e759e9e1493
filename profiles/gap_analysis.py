"""GPU busy fraction inside the timed steps of bench.py from a rocprofv3 kernel trace (union of kernel intervals / span).
usage: gap_analysis.py <kernel_trace.csv>   (rocprofv3 --kernel-trace --output-format csv -- python3 bench.py --steps 40 ...)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# window: from the 10th k_step launch to the last k_step launch
ks = [e for e in ev if "k_step" in e[2]]
t0, t1 = ks[10][0], ks[-1][0]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in ev:
    if e <= t0 or s >= t1:
        continue
    s, e = max(s, t0), min(e, t1)
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
steps = len(ks) - 11
print({"steps": steps, "span_ms_per_step": (t1 - t0) / steps / 1e6, "busy_ms_per_step": busy / steps / 1e6, "idle_fraction": 1 - busy / (t1 - t0)})
# overlap of the two conv chains: time with >= 2 conv kernels running / time with >= 1
pts = []
for s, e, n in ev:
    if "k_conv3x3" in n and s >= t0 and e <= t1:
        pts += [(s, 1), (e, -1)]
pts.sort()
depth, last, t_ge1, t_ge2 = 0, None, 0, 0
for t, d in pts:
    if last is not None:
        if depth >= 1:
            t_ge1 += t - last
        if depth >= 2:
            t_ge2 += t - last
    depth += d
    last = t
print({"conv_time_any_ms_per_step": t_ge1 / steps / 1e6, "conv_time_two_running_ms_per_step": t_ge2 / steps / 1e6})
