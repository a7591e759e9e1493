# does the trainer's side stream lose anything by sharing a hardware queue? (the evaluator binds seven launch-chain streams + the default = all eight queues)
O=gpurun_out
for q in 8 16 8 16; do
GPU_MAX_HW_QUEUES=$q python bench.py --playout 800 --steps 400 --warmup 16 --no-cpu-baseline --train-every 8 > $O/r05_v_trainer_q$q.json 2> $O/r05_v.err; echo "q$q rc $?"
python - "$q" <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/r05_v_trainer_q{sys.argv[1]}.json").read().splitlines() if l.startswith("{")][-1])
print("queues", sys.argv[1], round(d["value"]), round(d["ms_per_step"], 3), d["trainer_updates"])
PY
done
