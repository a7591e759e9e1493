O=gpurun_out
python -m pytest tests/test_gpu_host_mirror.py tests/test_gpu_00_bench_contract.py -q -k "record_shards or collect" > $O/r05_tests_S.log 2>&1; echo "tests rc $?"; tail -12 $O/r05_tests_S.log
python profiles/sink_microbench.py > $O/r05_sink_microbench.json 2> $O/r05_s.err; echo "rc $?"; cat $O/r05_sink_microbench.json; tail -3 $O/r05_s.err
