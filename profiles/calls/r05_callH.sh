# why is the N>1 path (group of one on RCCL) 26 % slower per tower layer? Hypothesis: hardware-queue aliasing (GPU_MAX_HW_QUEUES = 4 by default;
# the exchange's side stream and the process group's stream are created BEFORE the tower's three launch chains)
O=gpurun_out
set -e
GPU_MAX_HW_QUEUES=8 python bench.py --steps 400 --warmup 16 --no-cpu-baseline > $O/r05_h_plain_q8.json 2> $O/r05_h1.err; echo "plain q8"
GPU_MAX_HW_QUEUES=8 python bench.py --steps 400 --warmup 16 --no-cpu-baseline --rccl-group-of-one > $O/r05_h_rccl_q8.json 2> $O/r05_h2.err; echo "rccl q8"
python bench.py --steps 400 --warmup 16 --no-cpu-baseline --rccl-group-of-one --backend gloo > $O/r05_h_gloo_q4.json 2> $O/r05_h3.err; echo "gloo q4"
python bench.py --steps 400 --warmup 16 --no-cpu-baseline --rccl-group-of-one --exchange sync > $O/r05_h_rccl_sync_q4.json 2> $O/r05_h4.err; echo "rccl sync q4"
python - <<'PY'
import json
def L(f): return json.loads([l for l in open(f"gpurun_out/{f}").read().splitlines() if l.startswith("{")][-1])
for f in ("r05_h_plain_q8", "r05_h_rccl_q8", "r05_h_gloo_q4", "r05_h_rccl_sync_q4"):
    d = L(f + ".json"); print(f, round(d["value"]), round(d["ms_per_step"], 3), d["net_roofline"]["avg_launch_us"], d["net_roofline"]["avg_launch_us_full_batch"])
PY
