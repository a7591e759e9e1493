# flake hunt: the -m gpu suite twice in a row on one box
O=gpurun_out
for i in 1 2; do
python -m pytest tests -q -m gpu -p no:cacheprovider > $O/r05_gpu_tests_repeat_$i.log 2>&1; echo "run $i rc $?"; tail -2 $O/r05_gpu_tests_repeat_$i.log
done
