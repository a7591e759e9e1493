# Probe: would two independent half-batches (2 x 2048 boards, each its own stream of simulator + evaluator kernels) overlap better than
# one 4096-board lockstep batch? Crude stand-in: two bench.py PROCESSES of 2048 boards on the one GPU at the same time.
O=gpurun_out
python bench.py --boards 2048 --steps 2400 --warmup 16 --no-cpu-baseline --eval-cache-log2 23 > $O/r04_half_a.json 2>/dev/null &
PA=$!
python bench.py --boards 2048 --steps 2400 --warmup 16 --no-cpu-baseline --eval-cache-log2 23 > $O/r04_half_b.json 2>/dev/null &
PB=$!
wait $PA; wait $PB
python bench.py --boards 4096 --steps 1200 --warmup 16 --no-cpu-baseline > $O/r04_half_ref.json 2>/dev/null
python - <<'PY'
import json
def v(f):
    j=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1]); return j["value"], j["ms_per_step"]
a=v("gpurun_out/r04_half_a.json"); b=v("gpurun_out/r04_half_b.json"); r=v("gpurun_out/r04_half_ref.json")
print("two concurrent 2048-board processes:", round(a[0]), "+", round(b[0]), "=", round(a[0]+b[0]), "sims/s (ms/step", a[1], b[1], ") ; one 4096-board process:", round(r[0]), r[1])
PY
