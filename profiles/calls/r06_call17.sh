# round 6 call 17: the final code (head f8075698...) under load -- call 11 without the bounds-checked build (calls/r06_call16.sh)
O=gpurun_out
set -e
# (2) thirty moves of steady-state self-play with the evaluation cache in verify mode
python bench.py --steps 12000 --warmup 16 --no-cpu-baseline --cache-verify > $O/r06_bench_30moves_verify.json 2> $O/r06_bench_30moves_verify.err
# (3) one GPU's half of configs[4]: 4096 x 800
python bench.py --playout 800 --steps 800 --warmup 16 --no-cpu-baseline > $O/r06_cfg5_plain.json 2> $O/r06_cfg5_plain.err
# (4) the simulator alone under hipGraph replay (stub evaluator)
python bench.py --evaluator stub --graph --steps 1200 --warmup 16 --no-cpu-baseline > $O/r06_stub_evaluator_graph.json 2> $O/r06_stub.err
# (5) two and four ranks sharing the card over gloo (the N > 1 control flow with the asynchronous exchange)
python bench.py --gpus 2 --backend gloo --share-gpu --boards 1024 --steps 420 --warmup 8 --no-cpu-baseline > $O/r06_rehearsal_2ranks_gloo.json 2> $O/r06_reh2.err
python bench.py --gpus 4 --backend gloo --share-gpu --boards 1024 --steps 420 --warmup 8 --no-cpu-baseline > $O/r06_rehearsal_4ranks_gloo.json 2> $O/r06_reh4.err
python - <<'PY'
import json
def L(f): return json.loads([l for l in open(f"gpurun_out/{f}.json").read().splitlines() if l.startswith("{")][-1])
d = L("r06_bench_30moves_verify"); print("30 moves", round(d["value"]), round(d["ms_per_step"], 3), d["eval_cache"]["verify"], d["deviations"]["pruned_subtrees_total"], d["error_flags_any"])
d = L("r06_cfg5_plain"); print("4096x800", round(d["value"]), round(d["ms_per_step"], 3), d["engine_hbm_gb"])
d = L("r06_stub_evaluator_graph"); print("stub graph", round(d["value"]), round(d["ms_per_step"] * 1e3, 1), "us/step")
for f in ("r06_rehearsal_2ranks_gloo", "r06_rehearsal_4ranks_gloo"):
    d = L(f); m = d["multi_gpu"]; print(f, round(d["value"]), m["ranks_seen"], m["rank_step_ms"], m["collectives_in_window"], m["error_flags_any"], m["bad_records"])
PY
