set -e
O=gpurun_out
python -m pytest tests/test_gpu_conv.py tests/test_gpu_evaluator_depth.py tests/test_gpu_timed_path.py -m gpu -x -q > $O/stem_tests.log 2>&1 || (tail -30 $O/stem_tests.log; exit 1)
tail -2 $O/stem_tests.log
bash profiles/run_profile.sh r04 trace
python - <<'PY'
import csv, json
for r in csv.DictReader(open("profiles/r04_kernel_stats.csv")):
    if "conv3x3" in r["Name"] or "fc_" in r["Name"]:
        print(r["Name"][:70], r["Calls"], r["AverageNs"])
d = json.load(open("profiles/r04_summary.json"))["evaluator_tail"]
print({k: round(v["us_per_step"], 1) for k, v in d["kernels"].items()})
print(open("/tmp/prof_r04/trace_bench.json").read()[:200])
PY
