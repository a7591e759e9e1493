set -e
O=gpurun_out
python -m pytest tests -x -q -m gpu > $O/r04_gpu_tests.log 2>&1 || (tail -30 $O/r04_gpu_tests.log; exit 1)
tail -2 $O/r04_gpu_tests.log
python bench.py --steps 20 --warmup 5 > $O/r04_bench_20steps_v3.json 2> $O/r04_h1.err; echo done20
python bench.py --steps 1200 --warmup 16 --no-cpu-baseline > $O/r04_bench_3moves_v3.json 2> $O/r04_h2.err; echo done3
python - <<'PY'
import json
for f in ("r04_bench_20steps_v3", "r04_bench_3moves_v3"):
    d = json.load(open(f"gpurun_out/{f}.json")); print(f, round(d["value"]), d["ms_per_step"], d["roofline"]["frac"], d["net_roofline"]["frac"])
PY
