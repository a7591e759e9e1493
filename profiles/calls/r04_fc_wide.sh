set -e
O=gpurun_out
python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "fc_kernels or head_kernels" > $O/fcw_tests.log 2>&1 || (tail -30 $O/fcw_tests.log; exit 1)
tail -2 $O/fcw_tests.log
python profiles/fc_microbench.py --product > $O/r04_fc_microbench_wide.json 2> $O/fcw_mb.err
grep -v "ablation': 'torch" $O/fcw_mb.err | tail -30
