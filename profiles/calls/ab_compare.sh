# A/B of library builds on ONE box (device-to-device spread is larger than most kernel deltas):
#   build variants into build/diag (make -C chinesechesszero_amd/csrc ab NAME=x ABFLAGS=...), list them below, run via gpurun.
for rep in 1 2 3; do
for lib in "" libcczero_ab_encfirst.so; do
  for cfg in "4096 400 0" "4096 400 2"; do
    echo -n "${lib:-shipped} [$cfg]: "; CCZ_LIB=$lib timeout -k 10 200 python profiles/sim_microbench.py $cfg 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k_step', j['k_step_us(median,mean)'][0], 'select', j['k_select_us'][0], 'd_bar', round(j['d_bar'],2))"
  done
done
done
