# the bounds-checked build of the simulator kernels (no GPU sanitizer on this pool), built on the box, and the search / soak / harvest /
# cache / timed-path tests against it
O=gpurun_out
make -C chinesechesszero_amd/csrc bounds > $O/r05_bounds_build.log 2>&1; echo "build rc $?"
CCZ_LIB=libcczero_bounds.so python -m pytest tests/test_gpu_search.py tests/test_gpu_soak.py tests/test_gpu_harvest.py tests/test_gpu_eval_cache.py tests/test_gpu_timed_path.py tests/test_gpu_partition.py tests/test_gpu_rules_probe.py -q -m gpu > $O/r05_bounds_build_tests.log 2>&1; echo "tests rc $?"
tail -3 $O/r05_bounds_build_tests.log
