# round 5: how much does the concurrent trainer slow the rank it shares a GPU with? 4096 boards x 800 sims/move (configs[4], one GPU's half),
# with and without --train-every 8, interleaved on ONE box; then the cache-off and the simulator-only (stub evaluator) lines of configs[2]
set -e
O=gpurun_out
for i in 1 2; do
python bench.py --playout 800 --steps 400 --warmup 16 --no-cpu-baseline > $O/r05_cfg5_plain_$i.json 2> $O/r05_t1.err; echo "plain $i"
python bench.py --playout 800 --steps 400 --warmup 16 --no-cpu-baseline --train-every 8 > $O/r05_cfg5_trainer_$i.json 2> $O/r05_t2.err; echo "trainer $i"
done
python bench.py --steps 1200 --warmup 16 --no-cpu-baseline > $O/r05_3moves_cache_on.json 2> $O/r05_t3.err; echo "cache on"
python bench.py --steps 1200 --warmup 16 --no-cpu-baseline --eval-cache-log2 0 > $O/r05_3moves_cache_off.json 2> $O/r05_t4.err; echo "cache off"
python bench.py --steps 1200 --warmup 16 --no-cpu-baseline --evaluator stub > $O/r05_stub_evaluator.json 2> $O/r05_t5.err; echo "stub"
python - <<'PY'
import json
for f in ["cfg5_plain_1","cfg5_trainer_1","cfg5_plain_2","cfg5_trainer_2","3moves_cache_on","3moves_cache_off","stub_evaluator"]:
    d = json.loads(open(f"gpurun_out/r05_{f}.json").read().strip().splitlines()[-1]); print(f, round(d["value"]), round(d["ms_per_step"],3), d.get("trainer_updates"))
PY
