O=gpurun_out
python -m pytest tests/test_gpu_00_bench_contract.py -q -k "group_of_one" > $O/r05_tests_M.log 2>&1; echo "tests rc $?"; tail -15 $O/r05_tests_M.log
python bench.py --steps 12000 --warmup 16 --cache-verify --no-cpu-baseline > $O/r05_bench_30moves_verify.json 2> $O/r05_m1.err; echo "30 moves rc $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05_bench_30moves_verify.json").read().splitlines() if l.startswith("{")][-1])
print("30 moves", round(d["value"]), round(d["ms_per_step"], 3), d["moves_per_sec"], d["move_boundary"]["in_window"], d["move_boundary"]["games_finished"], d["eval_cache"]["verify"], d["eval_cache"]["fraction_of_needed_evaluations_skipped"], d["error_flags_any"])
PY
