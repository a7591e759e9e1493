O=gpurun_out
for cfg in "nccl 1" "nccl 0" "gloo 0"; do
python profiles/pg_slowdown_probe.py $cfg > $O/r05_pg_probe_$(echo $cfg | tr ' ' '_').json 2> $O/r05_g.err || tail -5 $O/r05_g.err
cat $O/r05_pg_probe_$(echo $cfg | tr ' ' '_').json
done
GPU_MAX_HW_QUEUES=8 python profiles/pg_slowdown_probe.py nccl 1 > $O/r05_pg_probe_nccl_1_hwq8.json 2>> $O/r05_g.err; cat $O/r05_pg_probe_nccl_1_hwq8.json
