# round 6 call 16: the bounds-checked build (no GPU ASan on this pool: every node-pool / path / record / chain index checked, CCZ_IDX) with the
# device-side loop of the one-game path (k_scouted_run) in it
O=gpurun_out
set -e
make -C chinesechesszero_amd/csrc bounds > /dev/null 2>&1
CCZ_LIB=libcczero_bounds.so timeout -k 10 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_soak.py tests/test_gpu_scouts.py tests/test_gpu_modes.py tests/test_gpu_ref_twins.py tests/test_gpu_frontends_parity.py -q -m gpu > $O/r06_bounds_build_tests.log 2>&1 || { tail -40 $O/r06_bounds_build_tests.log; exit 1; }
tail -2 $O/r06_bounds_build_tests.log
