# round 6 call 4: the whole -m gpu suite on the round's code so far (strict mode, tag compare in k_cache_plan, persistent kernel, ABI 7),
# then BASELINE configs[1] (1024 boards x 400): launch-chain sweep, one box, interleaved twice
O=gpurun_out
set -e
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/r06_gpu_tests_a.log 2>&1 || { tail -40 $O/r06_gpu_tests_a.log; exit 1; }
tail -2 $O/r06_gpu_tests_a.log
run() { # name, boards, env...
  name=$1; boards=$2; shift 2
  env "$@" python bench.py --boards $boards --steps 800 --warmup 16 --no-cpu-baseline > $O/r06_b${boards}_$name.json 2> $O/r06_b${boards}_$name.err
  python - <<PY
import json
d = json.loads([l for l in open("$O/r06_b${boards}_$name.json").read().splitlines() if l.startswith("{")][-1])
print("$boards $name", round(d["value"]), round(d["ms_per_step"], 3), round(d["net_roofline"]["avg_launch_us"], 2), round(d["net_roofline"]["frac"], 4))
PY
}
for rep in 1 2; do
  run chains2_$rep 1024 CCZ_TOWER_CHAINS=2
  run chains3_$rep 1024 CCZ_TOWER_CHAINS=3
  run chains4_$rep 1024 CCZ_TOWER_CHAINS=4
  run chains1_$rep 1024 CCZ_TOWER_CHAINS=1
  run pers256_chains2_$rep 1024 CCZ_CONV_PERSISTENT=256 CCZ_TOWER_CHAINS=2
done
run chains3 2048 CCZ_TOWER_CHAINS=3 CCZ_CONV_EDGE_TILES=0
run chains4 2048 CCZ_TOWER_CHAINS=4 CCZ_CONV_EDGE_TILES=0
run chains2 2048 CCZ_TOWER_CHAINS=2 CCZ_CONV_EDGE_TILES=0
