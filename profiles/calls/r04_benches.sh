set -e
O=gpurun_out
python bench.py --steps 20 --warmup 5 > $O/r04_bench_20steps.json 2> $O/r04_b1.err; echo "20 steps done"
python bench.py --steps 1200 --warmup 16 --no-cpu-baseline > $O/r04_bench_3moves.json 2> $O/r04_b2.err; echo "3 moves done"
CCZ_FUSED_HEADS=0 python bench.py --steps 1200 --warmup 16 --no-cpu-baseline > $O/r04_ab_3moves_torch_heads.json 2> $O/r04_b3.err; echo "3 moves torch heads done"
python bench.py --steps 20 --warmup 5 --cache-verify --no-cpu-baseline > $O/r04_bench_20steps_cache_verify.json 2> $O/r04_b4.err; echo "verify done"
python bench.py --boards 1024 --steps 400 --warmup 16 --no-cpu-baseline > $O/r04_bench_cfg2_1024boards.json 2> $O/r04_b5.err; echo "cfg2 done"
python bench.py --gpus 2 --backend gloo --share-gpu --boards 1024 --steps 40 --warmup 4 > $O/r04_rehearsal_2ranks_gloo.json 2> $O/r04_b6.err; echo "rehearsal done"
