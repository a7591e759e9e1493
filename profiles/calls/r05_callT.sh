O=gpurun_out
python -c "from chinesechesszero_amd import _lib; print('stale:', _lib.stale_build())"
python -m pytest tests -q -m gpu > $O/r05_gpu_tests_full.log 2>&1; echo "tests rc $?"
tail -4 $O/r05_gpu_tests_full.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/r05_smoke.log
python bench.py --steps 20 --warmup 5 > $O/r05_bench_20steps_last.json 2> $O/r05_t1.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05_bench_20steps_last.json").read().splitlines() if l.startswith("{")][-1])
print(round(d["value"]), d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["duration_source"][:40], d["roofline"]["code_hash"], d["roofline"]["profile_head"])
PY
