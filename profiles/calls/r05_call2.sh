# round 5: the new N>1 control flow on the one GPU (gloo rehearsals), then rank 0's calibrated half of configs[4]
O=gpurun_out
python -m pytest tests/test_gpu_00_bench_contract.py tests/test_gpu_rules_probe.py -q > $O/r05_contract_tests.log 2>&1; echo "tests rc $?"
tail -5 $O/r05_contract_tests.log
python bench.py --playout 800 --steps 400 --warmup 16 --no-cpu-baseline --train-every 8 --boards-rank0 auto > $O/r05_cfg5_trainer_auto.json 2> $O/r05_u1.err || { tail -30 $O/r05_u1.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_cfg5_trainer_auto.json").read().strip().splitlines()[-1])
print("auto", round(d["value"]), round(d["ms_per_step"], 3), d["trainer_updates"], d["config"]["boards_per_rank"], json.dumps(d["rank0_calibration"]))
PY
