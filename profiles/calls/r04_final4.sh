# final code of round 4: full -m gpu suite, full-size soak with the cache in verify mode, thirty moves of the bench's workload in verify mode,
# the driver-style line and three moves
set -e
O=gpurun_out
python -m pytest tests -x -q -m gpu > $O/r04_gpu_tests_final.log 2>&1 || (tail -30 $O/r04_gpu_tests_final.log; exit 1)
tail -2 $O/r04_gpu_tests_final.log
python profiles/selfplay_soak.py 4096 48 220 > $O/r04_selfplay_soak_4096_final.json 2> $O/r04_k1.err; echo "soak done"; tail -3 $O/r04_selfplay_soak_4096_final.json | cut -c1-400
python bench.py --steps 12000 --warmup 16 --cache-verify --no-cpu-baseline > $O/r04_bench_30moves_verify_final.json 2> $O/r04_k2.err; echo "30 moves done"
python bench.py --steps 20 --warmup 5 > $O/r04_bench_20steps_final.json 2> $O/r04_k3.err; echo done20
python bench.py --steps 1200 --warmup 16 --no-cpu-baseline > $O/r04_bench_3moves_final.json 2> $O/r04_k4.err; echo done3
python - <<'PY'
import json
for f in ("r04_bench_30moves_verify_final", "r04_bench_20steps_final", "r04_bench_3moves_final"):
    d = json.load(open(f"gpurun_out/{f}.json")); print(f, round(d["value"]), d["ms_per_step"], d["roofline"]["frac"], d["net_roofline"]["frac"], d["eval_cache"].get("verify"), d["error_flags_any"])
PY
