set -e
O=gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > $O/r04_smoke.log 2>&1; tail -1 $O/r04_smoke.log
python -m pytest tests -x -q -m gpu > $O/r04_gpu_suite.log 2>&1 || (tail -30 $O/r04_gpu_suite.log; exit 1)
tail -2 $O/r04_gpu_suite.log
python bench.py --steps 20 --warmup 5 > $O/r04_bench_20steps.json 2> $O/r04_f1.err; echo "20 steps done"
python bench.py --steps 3200 --warmup 16 --no-cpu-baseline > $O/r04_bench_8moves.json 2> $O/r04_f2.err; echo "8 moves done"
python profiles/selfplay_soak.py 4096 400 6 > $O/r04_selfplay_soak_4096_400sims.json 2> $O/r04_f3.err; echo "soak 400 done"
