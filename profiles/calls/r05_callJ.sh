O=gpurun_out
: > $O/r05_hwq_probe.txt
for rep in 1 2; do
for q in 4 8 16; do
for cfg in "0 none" "1 init_only" "1 exchange_first" "1 eval_first"; do
GPU_MAX_HW_QUEUES=$q python profiles/hwq_probe.py $cfg 2>> $O/r05_j.err | grep HWQ_PROBE >> $O/r05_hwq_probe.txt
done; done; done
cat $O/r05_hwq_probe.txt
