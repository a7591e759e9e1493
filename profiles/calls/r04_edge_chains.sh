# same box, interleaved: the default launch structure (one kernel per layer, two chains) against edge tiles with three chains
O=gpurun_out
for rep in 1 2 3; do
for cfg in "0 2" "1 3"; do
  set -- $cfg
  CCZ_CONV_EDGE_TILES=$1 CCZ_TOWER_CHAINS=$2 python bench.py --steps 1200 --warmup 16 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('edge=$1 chains=$2', round(j['value']), round(j['ms_per_step'],3), round(j['net_roofline']['avg_launch_us'],1), j['nodes_peak'])" | tee -a $O/r04_edge_chains.txt
done
done
