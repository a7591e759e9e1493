set -e
O=gpurun_out
python -m pytest tests/test_gpu_00_bench_contract.py tests/test_gpu_timed_path.py tests/test_gpu_evaluator_depth.py tests/test_gpu_eval_cache.py tests/test_gpu_full_configs.py tests/test_gpu_conv.py -x -q -m gpu > $O/r04_t6d.log 2>&1 || (tail -30 $O/r04_t6d.log; exit 1)
tail -2 $O/r04_t6d.log
python bench.py --steps 20 --warmup 5 > $O/r04_bench_20steps_v2.json 2> $O/r04_g1.err; echo done20
python bench.py --steps 1200 --warmup 16 --no-cpu-baseline > $O/r04_bench_3moves_v2.json 2> $O/r04_g2.err; echo done3
CCZ_CONV_EDGE_TILES=0 python bench.py --steps 1200 --warmup 16 --no-cpu-baseline > $O/r04_bench_3moves_v2_noedge.json 2> $O/r04_g3.err; echo done3b
