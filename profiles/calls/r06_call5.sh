# round 6 call 5: perft(6) kernels vs oracle (divide), the new GPU tests, BASELINE configs[1] with its new default (three launch chains)
O=gpurun_out
set -e
timeout -k 10 900 python profiles/perft6.py --workers 16 > $O/r06_perft6.json 2> $O/r06_perft6.err || { tail -20 $O/r06_perft6.err; exit 1; }
python - <<'PY'
import json; j = json.load(open("gpurun_out/r06_perft6.json")); print("perft6", j["perft_gpu"], j["perft6_oracle"], j["equal"], j["seconds_gpu"], j["seconds_oracle"])
PY
cp $O/r06_perft6.json profiles/r06_perft6.json
timeout -k 10 900 python -m pytest tests/test_gpu_rules.py tests/test_gpu_integration_snippet.py tests/test_gpu_conv.py tests/test_gpu_modes.py -x -q -m gpu > $O/r06_gpu_tests_b.log 2>&1 || { tail -40 $O/r06_gpu_tests_b.log; exit 1; }
tail -2 $O/r06_gpu_tests_b.log
python bench.py --boards 1024 --no-cpu-baseline > $O/r06_bench_cfg2_1024boards.json 2> $O/r06_bench_cfg2.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06_bench_cfg2_1024boards.json").read().splitlines() if l.startswith("{")][-1])
print("1024", round(d["value"]), round(d["ms_per_step"], 3), d["net_roofline"]["chains"], round(d["net_roofline"]["frac"], 4), d["roofline"]["frac"], d["deviations"])
PY
