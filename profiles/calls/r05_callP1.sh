# round 5, the library REBUILT from the final sources (the earlier calls of the round ran a libcczero.so built before the k_cache_plan edit:
# nothing had run make since -- see chinesechesszero_amd/_lib.py stale_build): the whole suite, smoke, then the profile passes
O=gpurun_out
python -c "from chinesechesszero_amd import _lib; print('stale:', _lib.stale_build(), 'source hash', _lib.source_hash())"
python -m pytest tests -q -m gpu > $O/r05_gpu_tests_full.log 2>&1; echo "tests rc $?"
tail -4 $O/r05_gpu_tests_full.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/r05_smoke.log
bash profiles/run_profile.sh r05 trace fetch write sq
