# round 5: tests that changed, then rank 0's self-calibrated half of configs[4] at two trainer cadences
O=gpurun_out
python -m pytest tests/test_gpu_00_bench_contract.py tests/test_gpu_rules_probe.py tests/test_gpu_partition.py tests/test_gpu_host_mirror.py tests/test_gpu_frontends.py -q > $O/r05_tests3.log 2>&1; echo "tests rc $?"
tail -5 $O/r05_tests3.log
for te in 8 16; do
python bench.py --playout 800 --steps 400 --warmup 16 --no-cpu-baseline --train-every $te --boards-rank0 auto > $O/r05_cfg5_trainer_auto_te$te.json 2> $O/r05_u$te.err || { tail -30 $O/r05_u$te.err; exit 1; }
done
python - <<'PY'
import json
for te in (8, 16):
    d = json.loads(open(f"gpurun_out/r05_cfg5_trainer_auto_te{te}.json").read().strip().splitlines()[-1])
    print("auto te", te, round(d["value"]), round(d["ms_per_step"], 3), d["trainer_updates"], d["config"]["boards_per_rank"], json.dumps({k: v for k, v in d["rank0_calibration"].items() if k != "what"}))
PY
