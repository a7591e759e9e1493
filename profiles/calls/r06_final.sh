# round 6, final code: the whole -m gpu suite, smoke(), the driver's bench shapes -- ONE box, one call
O=gpurun_out
python -c "from chinesechesszero_amd import _lib, build; print('stale:', _lib.stale_build(), 'source hash', _lib.source_hash(), 'code hash', build.code_hash())"
set -e
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/r06_gpu_tests_full.log 2>&1 || { tail -40 $O/r06_gpu_tests_full.log; exit 1; }
tail -2 $O/r06_gpu_tests_full.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.log 2>&1; tail -1 $O/r06_smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_20steps.json 2> $O/r06_bench_20steps.err
python bench.py --steps 1200 --warmup 16 --no-cpu-baseline > $O/r06_bench_3moves.json 2> $O/r06_bench_3moves.err
python bench.py > $O/r06_bench_default.json 2> $O/r06_bench_default.err
python - <<'PY'
import json
for f in ("r06_bench_20steps", "r06_bench_3moves", "r06_bench_default"):
    d = json.loads([l for l in open(f"gpurun_out/{f}.json").read().splitlines() if l.startswith("{")][-1])
    r = d["roofline"]
    print(f, round(d["value"]), round(d["ms_per_step"], 3), round(d["moves_per_sec"], 1), "frac", round(r["frac"], 4), round(r["avg_launch_us"], 1), "net", round(d["net_roofline"]["frac"], 4),
          "profile", r["frac_at_committed_rocprofv3_duration"], (r["committed_profile_note"] or "")[:60], d["deviations"]["pruned_subtrees_in_window"], d["deviations"]["truncated_games_in_window"], (d.get("cpu_baseline") or {}).get("value"))
PY
