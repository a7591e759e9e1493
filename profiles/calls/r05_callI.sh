# round 5: after the hardware-queue fix (package default GPU_MAX_HW_QUEUES=8): the full-size window plain and with the N>1 path in a group of one on RCCL,
# asynchronous and blocking exchange, one box, interleaved; then the new test
O=gpurun_out
set -e
python bench.py --steps 1200 --warmup 16 --no-cpu-baseline > $O/r05_i_plain.json 2> $O/r05_i1.err; echo "plain"
python bench.py --steps 1200 --warmup 16 --no-cpu-baseline --rccl-group-of-one > $O/r05_i_rccl_async.json 2> $O/r05_i2.err; echo "rccl async"
python bench.py --steps 1200 --warmup 16 --no-cpu-baseline --rccl-group-of-one --exchange sync > $O/r05_i_rccl_sync.json 2> $O/r05_i3.err; echo "rccl sync"
python bench.py --steps 20 --warmup 5 > $O/r05_i_20steps.json 2> $O/r05_i4.err; echo "20 steps"
python - <<'PY'
import json
def L(f): return json.loads([l for l in open(f"gpurun_out/{f}").read().splitlines() if l.startswith("{")][-1])
for f in ("r05_i_plain", "r05_i_rccl_async", "r05_i_rccl_sync", "r05_i_20steps"):
    d = L(f + ".json"); m = d.get("multi_gpu", {})
    print(f, round(d["value"]), round(d["ms_per_step"], 3), d["net_roofline"]["avg_launch_us"], d["gpu_max_hw_queues"], d["roofline"]["frac"], {k: m.get(k) for k in ("collectives_in_window", "exchange_host_ms_rank0", "exchange_max_call_ms_per_rank", "bytes_sent_rank0", "drain_ms_rank0", "gather_ms")})
PY
python -m pytest tests/test_gpu_00_bench_contract.py -q -s -k "launch_chains" 2>&1 | tail -5
