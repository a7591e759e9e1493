# the driver's shapes on the final code: no flags; and its N = 1 scaling command under the launcher
O=gpurun_out
( time python bench.py > $O/r05_bench_default.json 2> $O/r05_q1.err ) 2> $O/r05_q1.time; echo "default rc $?"; tail -3 $O/r05_q1.time
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_launcher_n1.json 2> $O/r05_q2.err; echo "launcher rc $?"
python - <<'PY'
import json
def L(f): return json.loads([l for l in open(f"gpurun_out/{f}").read().splitlines() if l.startswith("{")][-1])
for f in ("r05_bench_default", "r05_bench_launcher_n1"):
    d = L(f + ".json"); print(f, round(d["value"]), round(d["ms_per_step"], 3), d["steps"], d["warmup"], d["roofline"]["frac"], d["roofline"]["duration_source"][:50], "cpu", d.get("cpu_baseline", {}).get("value"), "multi_gpu" in d)
PY
