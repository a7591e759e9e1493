# round 6 call 3: is there headroom from perfect fill? Evaluation cache OFF (all 4096 rows live = 1280 tiles = exactly 5 per persistent
# workgroup, no quantisation): shipped launch structure against ONE persistent launch per layer, one box, interleaved
O=gpurun_out
set -e
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 1200 --warmup 16 --no-cpu-baseline --eval-cache-log2 0 > $O/r06_nocache_$name.json 2> $O/r06_nocache_$name.err
  python - <<PY
import json
d = json.loads([l for l in open("$O/r06_nocache_$name.json").read().splitlines() if l.startswith("{")][-1])
print("$name", round(d["value"]), round(d["ms_per_step"], 3), d["net_roofline"]["avg_launch_us"], d["net_roofline"]["frac"])
PY
}
run shipped_1 CCZ_NOP=1
run pers256_1chain CCZ_CONV_PERSISTENT=256 CCZ_CONV_EDGE_TILES=0 CCZ_TOWER_CHAINS=1
run shipped_2 CCZ_NOP=1
run pers256_1chain_2 CCZ_CONV_PERSISTENT=256 CCZ_CONV_EDGE_TILES=0 CCZ_TOWER_CHAINS=1
run pers256_2chains CCZ_CONV_PERSISTENT=256 CCZ_CONV_EDGE_TILES=0 CCZ_TOWER_CHAINS=2
