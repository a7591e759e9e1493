# round 5: the tower's counter passes (merged into profiles/r05_summary.json), then the bench lines of the final code on ONE box
O=gpurun_out
bash profiles/run_profile.sh r05 cfetch cwrite csq cmfma || echo "profile passes failed"
set -e
python bench.py --steps 20 --warmup 5 > $O/r05_bench_20steps.json 2> $O/r05_b1.err; echo "20 steps"
python bench.py --steps 1200 --warmup 16 --no-cpu-baseline > $O/r05_bench_3moves.json 2> $O/r05_b2.err; echo "3 moves"
python bench.py --steps 1200 --warmup 16 --no-cpu-baseline --evaluator stub --graph > $O/r05_stub_evaluator_graph.json 2> $O/r05_b3.err; echo "stub graph"
python bench.py --boards 1024 --steps 400 --warmup 16 --no-cpu-baseline > $O/r05_bench_cfg2_1024boards.json 2> $O/r05_b4.err; echo "cfg2"
python bench.py --gpus 2 --backend gloo --share-gpu --boards 1024 --steps 40 --warmup 4 > $O/r05_rehearsal_2ranks_gloo.json 2> $O/r05_b5.err; echo "2 ranks"
python bench.py --gpus 4 --backend gloo --share-gpu --boards 1024 --steps 40 --warmup 4 > $O/r05_rehearsal_4ranks_gloo.json 2> $O/r05_b6.err; echo "4 ranks"
python - <<'PY'
import json
def L(f): return json.loads([l for l in open(f"gpurun_out/{f}").read().splitlines() if l.startswith("{")][-1])
d = L("r05_bench_20steps.json"); r = d["roofline"]; print("20 steps", round(d["value"]), round(d["ms_per_step"], 3), "frac", round(r["frac"], 4), r["avg_launch_us"], "raw", r["avg_launch_us_hip_events_raw"], r["duration_source"][:60], "net", d["net_roofline"]["frac"])
d = L("r05_bench_3moves.json"); print("3 moves", round(d["value"]), round(d["ms_per_step"], 3), d["roofline"]["frac"])
d = L("r05_stub_evaluator_graph.json"); print("stub graph", round(d["value"]), d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
d = L("r05_bench_cfg2_1024boards.json"); print("1024", round(d["value"]), d["ms_per_step"])
for f in ("r05_rehearsal_2ranks_gloo.json", "r05_rehearsal_4ranks_gloo.json"):
    d = L(f); m = d["multi_gpu"]; print(f, round(d["value"]), m["ranks_seen"], m["rank_step_ms"], m["collectives_in_window"], m["collectives_in_drain"], m["exchange_max_call_ms_per_rank"], m["error_flags_any"])
PY
