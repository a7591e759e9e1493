# rocm-smi sampled every ~5 s while bench.py runs three moves (final round-4 code: edge-pair tower kernel, three chains, hand-written tail)
O=gpurun_out
python bench.py --steps 3000 --warmup 16 --no-cpu-baseline > $O/r04_power_bench.json 2>/dev/null &
PB=$!
sleep 45
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showtemp --showclocks --showpower 2>/dev/null | grep -E "junction|sclk|Package Power" >> $O/r04_power_clock_raw.txt
  echo "---" >> $O/r04_power_clock_raw.txt
  sleep 5
done
wait $PB
python -c "
import json; j=json.loads([l for l in open('$O/r04_power_bench.json').read().splitlines() if l.startswith('{')][-1]); print('# bench line of that run:', round(j['value']), 'sims/s,', round(j['ms_per_step'],2), 'ms/step')" >> $O/r04_power_clock_raw.txt
