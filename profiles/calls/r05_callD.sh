O=gpurun_out
python -m pytest tests/test_gpu_00_bench_contract.py tests/test_gpu_host_mirror.py -q > $O/r05_tests_D.log 2>&1; echo "tests rc $?"
tail -15 $O/r05_tests_D.log
python bench.py --gpus 2 --backend gloo --share-gpu --boards 1024 --steps 40 --warmup 4 > $O/r05_rehearsal_2ranks_gloo.json 2> $O/r05_d1.err; echo "2 ranks rc $?"
python bench.py --gpus 4 --backend gloo --share-gpu --boards 1024 --steps 40 --warmup 4 > $O/r05_rehearsal_4ranks_gloo.json 2> $O/r05_d2.err; echo "4 ranks rc $?"
