# round 5: the whole -m gpu suite and smoke() on the final code, one box; then the driver's command once more (tick_until in the loop)
O=gpurun_out
python -m pytest tests -q -m gpu -x > $O/r05_gpu_tests_full.log 2>&1; echo "tests rc $?"
tail -4 $O/r05_gpu_tests_full.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/r05_smoke.log
python bench.py --gpus 2 --backend gloo --share-gpu --boards 1024 --steps 40 --warmup 4 > $O/r05_rehearsal_2ranks_gloo_v2.json 2> $O/r05_c1.err; echo "2 ranks rc $?"
