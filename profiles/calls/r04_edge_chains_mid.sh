O=gpurun_out
for b in 2048 3072; do
for cfg in "0 2" "1 3" "0 3" "1 2"; do
  set -- $cfg
  CCZ_CONV_EDGE_TILES=$1 CCZ_TOWER_CHAINS=$2 python bench.py --boards $b --steps 800 --warmup 16 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$b boards edge=$1 chains=$2', round(j['value']), round(j['ms_per_step'],3), j['nodes_peak'])" | tee -a $O/r04_edge_chains_1024.txt
done
done
