# round 6 call 8: the host twins (oracle/ccz_ref.c) against the library through one driver; the driver-style bench once more (the committed
# profile is now of this code: frac_at_committed_rocprofv3_duration must appear)
O=gpurun_out
set -e
timeout -k 10 600 python -m pytest tests/test_gpu_ref_twins.py -x -q -m gpu > $O/r06_twins.log 2>&1 || { tail -60 $O/r06_twins.log; exit 1; }
tail -2 $O/r06_twins.log
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r06_bench_20steps_b.json 2> $O/r06_bench_20steps_b.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06_bench_20steps_b.json").read().splitlines() if l.startswith("{")][-1]); r = d["roofline"]
print(round(d["value"]), r["frac"], r["frac_at_committed_rocprofv3_duration"], r["live_over_profile"], r["committed_profile_note"], r["traffic"])
PY
