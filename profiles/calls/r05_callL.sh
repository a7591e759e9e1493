# round 5, after the hardware-queue fix: (1) the trainer rank again (its side stream is one more stream next to the tower's chains), (2) 100 move boundaries
# through the asynchronous exchange on RCCL in a group of one, (3) the whole -m gpu suite + smoke on the final code
O=gpurun_out
python bench.py --playout 800 --steps 400 --warmup 16 --no-cpu-baseline > $O/r05_cfg5_plain_q8.json 2> $O/r05_l1.err; echo "plain rc $?"
python bench.py --playout 800 --steps 400 --warmup 16 --no-cpu-baseline --train-every 8 > $O/r05_cfg5_trainer_q8.json 2> $O/r05_l2.err; echo "trainer rc $?"
python bench.py --rccl-group-of-one --boards 512 --blocks 2 --channels 256 --playout 32 --max-plies 40 --preroll-plies 40 --steps 3200 --warmup 4 --gather-plies 4096 --no-cpu-baseline > $O/r05_soak_rccl_group_of_one.json 2> $O/r05_l3.err; echo "rccl soak rc $?"
python - <<'PY'
import json
def L(f): return json.loads([l for l in open(f"gpurun_out/{f}").read().splitlines() if l.startswith("{")][-1])
for f in ("r05_cfg5_plain_q8", "r05_cfg5_trainer_q8"):
    d = L(f + ".json"); print(f, round(d["value"]), round(d["ms_per_step"], 3), d["trainer_updates"], d["gpu_max_hw_queues"])
d = L("r05_soak_rccl_group_of_one.json"); m = d["multi_gpu"]
print("rccl soak", round(d["value"]), d["ms_per_step"], d["move_boundary"]["in_window"], {k: m.get(k) for k in ("backend", "collectives_in_window", "collectives_in_drain", "exchanges_decided", "exchanges_without_a_collective", "exchange_host_ms_rank0", "exchange_max_call_ms_per_rank", "rows_gathered", "games_gathered", "replay_rows_total", "bad_records", "error_flags_any")})
PY
python -m pytest tests -q -m gpu > $O/r05_gpu_tests_full.log 2>&1; echo "tests rc $?"
tail -4 $O/r05_gpu_tests_full.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/r05_smoke.log
