# round 6 call 9: configs[1] (1024 boards): the remaining launch-structure knobs, one box, interleaved
O=gpurun_out
set -e
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --boards 1024 --steps 800 --warmup 16 --no-cpu-baseline > $O/r06_k1024_$name.json 2> $O/r06_k1024_$name.err
  python - <<PY
import json
d = json.loads([l for l in open("$O/r06_k1024_$name.json").read().splitlines() if l.startswith("{")][-1])
print("1024 $name", round(d["value"]), round(d["ms_per_step"], 3), round(d["net_roofline"]["avg_launch_us"], 2), round(d["net_roofline"]["frac"], 4), d["eval_cache"]["rows_computed_per_step"])
PY
}
for rep in 1 2; do
  run default_$rep CCZ_NOP=1
  run nozigzag_$rep CCZ_CONV_ZIGZAG=0
  run groups2_$rep CCZ_TOWER_GROUPS=2 CCZ_TOWER_CHAINS=2
  run nhwc_$rep CCZ_CONV_LAYOUT=nhwc
  run nofusedlast_$rep CCZ_FUSED_LAST=0
done
