# round 6 call 2: persistent tower kernel in the workload (4096 boards x 400, three moves), one box, interleaved
O=gpurun_out
set -e
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 1200 --warmup 16 --no-cpu-baseline > $O/r06_ab_$name.json 2> $O/r06_ab_$name.err
  python - <<PY
import json
d = json.loads([l for l in open("$O/r06_ab_$name.json").read().splitlines() if l.startswith("{")][-1])
print("$name", round(d["value"]), round(d["ms_per_step"], 3), d["net_roofline"]["avg_launch_us"], d["net_roofline"]["frac"])
PY
}
run shipped_1 CCZ_NOP=1
run pers256_2chains CCZ_CONV_PERSISTENT=256 CCZ_CONV_EDGE_TILES=0 CCZ_TOWER_CHAINS=2
run plain_2chains CCZ_CONV_EDGE_TILES=0 CCZ_TOWER_CHAINS=2
run pers256_1chain CCZ_CONV_PERSISTENT=256 CCZ_CONV_EDGE_TILES=0 CCZ_TOWER_CHAINS=1
run pers256_edge_3chains CCZ_CONV_PERSISTENT=256 CCZ_CONV_EDGE_TILES=1 CCZ_TOWER_CHAINS=3
run pers256_3chains CCZ_CONV_PERSISTENT=256 CCZ_CONV_EDGE_TILES=0 CCZ_TOWER_CHAINS=3
run shipped_2 CCZ_NOP=1
