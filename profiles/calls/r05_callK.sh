O=gpurun_out
: > $O/r05_hwq_probe_after.txt
for q in 4 8; do
for cfg in "0 none" "1 init_only" "1 exchange_first" "1 eval_first" "1 net_then_exchange"; do
GPU_MAX_HW_QUEUES=$q python profiles/hwq_probe.py $cfg 2>> $O/r05_k.err | grep HWQ_PROBE >> $O/r05_hwq_probe_after.txt
done; done
cat $O/r05_hwq_probe_after.txt
python -m pytest tests/test_gpu_00_bench_contract.py -q -s -k "launch_chains" 2>&1 | tail -5
