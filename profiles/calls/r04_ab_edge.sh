set -e
O=gpurun_out
python -m pytest tests/test_gpu_conv.py tests/test_gpu_evaluator_depth.py tests/test_gpu_timed_path.py tests/test_gpu_eval_cache.py -x -q -m gpu > $O/r04_t6b.log 2>&1 || (tail -30 $O/r04_t6b.log; exit 1)
tail -2 $O/r04_t6b.log
for rep in 1 2; do
CCZ_CONV_EDGE_TILES=1 python bench.py --steps 1200 --warmup 16 --no-cpu-baseline > $O/r04_ab_3moves_edge_tiles_$rep.json 2> $O/r04_e1.err; echo "edge tiles $rep done"
python bench.py --steps 1200 --warmup 16 --no-cpu-baseline > $O/r04_ab_3moves_one_kernel_$rep.json 2> $O/r04_e2.err; echo "one kernel $rep done"
done
